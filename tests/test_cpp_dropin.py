"""The C++ drop-in headers (libstevi_amd/include): they compile and link against libstevi_hip.so on CPU, and on the
GPU the reference benchmark's call chain written with the reference's names matches the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_chain.cpp")


def build(tmp_path):
    exe = str(tmp_path / "dropin_chain")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "libstevi_amd", "include"), SRC, "-o", exe,
           "-L", os.path.join(ROOT, "libstevi_amd"), "-lstevi_hip", "-Wl,-rpath," + os.path.join(ROOT, "libstevi_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    subprocess.check_call(cmd)
    return exe


def test_headers_compile_and_link(tmp_path):
    exe = build(tmp_path)
    assert os.path.exists(exe)


def test_multidim_array_members_of_the_reference_call_sites(tmp_path):
    """tests/cpp/multidim_compat.cpp uses subView / sliceView / indexDimView / buildReshapedView / takePointer / IndexBlock::setZero,
    moveToNextIndex / IndexConverter / ConstView the way the cited reference lines do, and checks what the shims hand to the C ABI for
    strided and ConstView arrays; host only (no GPU call), under AddressSanitizer + UBSan."""
    exe = str(tmp_path / "multidim_compat")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "libstevi_amd", "include"), os.path.join(ROOT, "tests", "cpp", "multidim_compat.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "libstevi_amd"), "-lstevi_hip", "-Wl,-rpath," + os.path.join(ROOT, "libstevi_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_reference_call_chain_matches_oracle(tmp_path):
    import oracle as so
    from helpers import parallax_pair, refined_2d_mismatch
    exe = build(tmp_path)
    src, tgt, _ = parallax_pair(40, 64, 12, 10, 20, 2, 7, seed=21)
    H, W, D = src.shape[0], src.shape[1], 24
    tgt.tofile(tmp_path / "l.f32")
    src.tofile(tmp_path / "r.f32")
    out = subprocess.run([exe, str(H), str(W), str(D), str(tmp_path / "l.f32"), str(tmp_path / "r.f32"), str(tmp_path / "o")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
    vol = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    got_vol = np.fromfile(tmp_path / "o_census_sgm.f32", np.float32).reshape(H, W, D)
    assert np.array_equal(got_vol.view(np.uint32), vol.view(np.uint32))
    got_disp = np.fromfile(tmp_path / "o_census_disp.i32", np.int32).reshape(H, W)
    assert np.array_equal(got_disp, so.index_to_disp(so.extract_index(vol, so.COST)))
    # the same chain on HipBridge::DeviceArray (volumes never leave the GPU) gives the same bits
    assert np.array_equal(np.fromfile(tmp_path / "o_census_disp_dev.i32", np.int32).reshape(H, W), got_disp)
    # ... with or without the library's statement about the volume's contents (dropped by a mutable access)
    assert np.array_equal(np.fromfile(tmp_path / "o_census_disp_dev_touched.i32", np.int32).reshape(H, W), got_disp)
    # ... also when the volume was made by a thread that has exited since (the array is freed by device, not through that thread's context)
    assert np.array_equal(np.fromfile(tmp_path / "o_census_disp_handed.i32", np.int32).reshape(H, W), got_disp)
    a, b = np.fromfile(tmp_path / "o_census_ref_dev.f32", np.float32), np.fromfile(tmp_path / "o_census_ref_host.f32", np.float32)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    cv2 = so.unfold_cost_volume_2d(so.ZNCC, tgt, src, 2, 2, (-1, 2), (-2, 3))
    got2 = np.fromfile(tmp_path / "o_zncc2d_cv.f32", np.float32).reshape(cv2.shape)
    assert np.array_equal(np.isnan(got2), np.isnan(cv2)) and np.nanmax(np.abs(got2 - cv2)) <= 1e-4
    got2f = np.fromfile(tmp_path / "o_zncc2d_cv_feat.f32", np.float32).reshape(cv2.shape)
    assert np.nanmax(np.abs(got2f - cv2)) <= 1e-4  # same volume through unfold + featureVolume2CostVolume(searchOffset<2>)
    disp2 = np.fromfile(tmp_path / "o_zncc2d_disp.i32", np.int32).reshape(H, W, 2)
    assert np.array_equal(disp2, so.index_2d_to_disp(so.extract_index_2d(got2, so.SCORE), -1, -2))
    idx2 = so.extract_index_2d(got2, so.SCORE)
    tcv2 = so.truncated_bidirectional_cv(got2, idx2, 1, 1)
    for name, exp in (("iso", so.refine_disp_2d(tcv2, disp2, so.EQUIANGULAR, so.ISOTROPIC)), ("aniso", so.refine_disp_2d(tcv2, disp2, so.PARABOLA, so.ANISOTROPIC)),
                      ("patch", so.refine_disp_2d_patch(tcv2, disp2, so.PARABOLA))):
        got = np.fromfile(tmp_path / f"o_zncc2d_ref_{name}.f32", np.float32).reshape(H, W, 2)
        bad, flipped = refined_2d_mismatch(got, exp, disp2)
        assert bad == 0.0 and flipped <= 0.01, (name, bad, flipped)
    etcv, edisp = so.hierarchical_truncated_cv(so.ZNCC, 2, tgt, src, 2, 2, D, 2)
    hdisp = np.fromfile(tmp_path / "o_hier_disp.i32", np.int32).reshape(H, W)
    same = hdisp == edisp
    assert same.mean() >= 0.995  # default (register-blocked) coarsest volume: see tests/test_gpu_hierarchical.py
    assert np.max(np.abs(np.fromfile(tmp_path / "o_hier_tcv.f32", np.float32).reshape(H, W, 5)[same] - etcv[same])) <= 1e-4
    half = so.average_pooling_downsample(src, 2)
    assert np.array_equal(np.fromfile(tmp_path / "o_half.f32", np.float32).reshape(half.shape), half)
    feats = so.unfold(src, 1, 1)
    assert np.array_equal(np.fromfile(tmp_path / "o_mean.f32", np.float32).reshape(H, W), so.channels_mean(feats))
    assert np.array_equal(np.fromfile(tmp_path / "o_zm.f32", np.float32).reshape(feats.shape), so.affine_feature_volume(feats, so.channels_mean(feats)))
    assert np.array_equal(np.fromfile(tmp_path / "o_zncc_feat.f32", np.float32).reshape(feats.shape), so.feature_volume_for_match_func(so.ZNCC, feats))
    assert np.array_equal(np.fromfile(tmp_path / "o_words.u32", np.uint32).reshape(H, W, 2), so.census_transform(src, 3, 3))
    import libstevi_amd as sv
    mask = sv.CompressorGenerators.GrPix17R3Filter()
    ccv = so.feature_cost_volume(so.ZNCC, so.unfold_compressed(tgt, mask), so.unfold_compressed(src, mask), D)
    got_c = np.fromfile(tmp_path / "o_compressed_cv.f32", np.float32).reshape(H, W, D)
    assert np.array_equal(np.isnan(got_c), np.isnan(ccv)) and np.nanmax(np.abs(got_c - ccv)) <= 1e-4
    pm, _ = so.cacheless_patch_match(so.ZNCC, 2, src[:, :, None], tgt[:, :, None], 2, 2, ((-2, 2), (0, 12)), 6, 4, seed=99)
    got_pm = np.fromfile(tmp_path / "o_pm_disp.i32", np.int32).reshape(H, W, 2)
    assert np.array_equal(got_pm, pm)
    pm_tcv = so.on_demand_truncated_cv(so.ZNCC, src[:, :, None], tgt[:, :, None], 2, 2, ((-2, 2), (0, 12)), pm, 1)
    pm_ref = so.refine_disp_2d(pm_tcv, pm, so.EQUIANGULAR, so.ISOTROPIC)
    bad, flipped = refined_2d_mismatch(np.fromfile(tmp_path / "o_pm_refined.f32", np.float32).reshape(H, W, 2), pm_ref, pm)
    assert bad == 0.0 and flipped <= 0.01
    # patchMatch on unfolded images with a NumbersCache / an initializer callback (benchmarkStereoMatchingModels.cpp:176-206)
    fl, fr = so.unfold(tgt, 2, 2), so.unfold(src, 2, 2)
    assert np.array_equal(np.fromfile(tmp_path / "o_pmf_fvol_left.f32", np.float32).reshape(fl.shape), fl)
    pmf, _ = so.patch_match(so.ZNCC, 1, fr, fl, (0, 12), 5, 4, seed=77)
    assert np.array_equal(np.fromfile(tmp_path / "o_pmf_zncc.i32", np.int32).reshape(H, W, 1), pmf)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    init = ((ii + 2 * jj) % 13).astype(np.int32)[:, :, None]
    pmi, _ = so.patch_match(so.SAD, 1, fr, fl, (0, 12), 4, 3, seed=77, init=init)
    assert np.array_equal(np.fromfile(tmp_path / "o_pmf_sad_init.i32", np.int32).reshape(H, W, 1), pmi)
    ncc = so.unfold_cost_volume(so.NCC, tgt, src, 4, 4, D)
    got_ncc = np.fromfile(tmp_path / "o_ncc_cv.f32", np.float32).reshape(H, W, D)
    assert np.max(np.abs(got_ncc - ncc)) <= 1e-4
    svol = so.sgm(got_ncc, 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    idx = so.extract_index(svol, so.SCORE)
    assert np.array_equal(np.fromfile(tmp_path / "o_ncc_idx.i32", np.int32).reshape(H, W), idx)
    ref = so.refine_disp(so.truncated_cost_volume(svol, idx, 4, 4, 1), idx, so.PARABOLA)
    got_ref = np.fromfile(tmp_path / "o_ncc_refined.f32", np.float32).reshape(H, W)
    assert np.array_equal(np.isnan(got_ref), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.max(np.abs(got_ref[ok] - ref[ok])) <= 1e-4
    # uint8 images through the same headers (widened on the device; equal to the float32 results of the same values)
    t8 = np.fromfile(tmp_path / "o_u8_target.u8", np.uint8).reshape(H, W)
    s8 = np.fromfile(tmp_path / "o_u8_source.u8", np.uint8).reshape(H, W)
    assert np.array_equal(t8, ((tgt + 1.0) * np.float32(127.5)).astype(np.uint8))
    sad8 = so.unfold_cost_volume(so.SAD, t8.astype(np.float32), s8.astype(np.float32), 2, 2, D)
    assert np.array_equal(np.fromfile(tmp_path / "o_u8_sad.f32", np.float32).reshape(H, W, D), sad8)
    assert np.array_equal(np.fromfile(tmp_path / "o_u8_words.u32", np.uint32).reshape(H, W, 2), so.census_transform(s8.astype(np.float32), 3, 3))
    unf8 = so.unfold(s8.astype(np.float32), 1, 2)
    assert np.array_equal(np.fromfile(tmp_path / "o_u8_unfold.u8", np.uint8).reshape(unf8.shape), unf8.astype(np.uint8))


def build_sharded(tmp_path):
    exe = str(tmp_path / "bench_sharded")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "libstevi_amd", "include"), "-I", "/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tools", "bench_sharded.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "libstevi_amd"), "-lstevi_hip", "-Wl,-rpath," + os.path.join(ROOT, "libstevi_amd"),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lrccl", "-lamdhip64", "-lpthread"])
    return exe


def test_sharded_cpp_host_compiles_links_and_never_oversubscribes(tmp_path):
    """The C++ multi-GPU host (correlation/sharded.h + svh_census_exchange_keys, one process per GPU, RCCL) compiles and links on the CPU;
    asked for more ranks than there are GPUs it says so and exits cleanly."""
    import json
    exe = build_sharded(tmp_path)
    out = subprocess.run([exe, "--rank", "0", "--world", "64", "--id-file", str(tmp_path / "id")], capture_output=True, text=True)
    assert out.returncode == 0 and "skipped" in json.loads(out.stdout)


@pytest.mark.gpu
def test_sharded_cpp_host_one_rank_through_rccl(tmp_path):
    """One rank, the all-reduce forced: svh_census_exchange_keys runs on a real RCCL communicator (both the plane-0 form and, with a
    narrower target, the two-plane form are covered by the geometry) and the map equals the unsharded call's."""
    import json
    exe = build_sharded(tmp_path)
    out = subprocess.run([exe, "--rank", "0", "--world", "1", "--id-file", str(tmp_path / "id"), "--width", "640", "--height", "96", "--disparities", "128",
                          "--steps", "2", "--exchange-always", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["pixels_differing_from_one_gpu"] == 0 and line["n_gpus"] == 1
