"""Reads the per-row barrier stamps of the fused Score-branch sweep (svh_sgm_sweep.hip, sgm_score_down_kernel; written when the
environment variable SVH_SGM_DOWN_STAMPS names a file) and prints, for a few strips, when they started and ended, their time per
row at the start / middle / end of their life, the lag between neighbouring strips at a few rows and the number of strips alive
at a few instants.  100 MHz ticks -> microseconds.
  SVH_SGM_DOWN_STAMPS=/tmp/stamps.bin python tools/sgm_score_probe.py && python tools/sgm_down_stamps_report.py /tmp/stamps.bin"""
import numpy as np, sys
f=open(sys.argv[1],'rb'); hdr=np.frombuffer(f.read(16),np.int32); ns,H,W,WB=hdr
st=np.frombuffer(f.read(),np.uint64).reshape(ns,H+1).astype(np.float64)
t0=st[st>0].min()
st=np.where(st>0,(st-t0)/100.0,np.nan)  # us
print("strips",ns,"H",H,"total us",np.nanmax(st))
for s in [0,1,2,5,20,50,100,150,200,255,256,257,300,350,380,390]:
    if s>=ns: continue
    r=st[s]; idx=np.where(~np.isnan(r))[0]
    if len(idx)<3: print(s,"rows",len(idx)); continue
    d=np.diff(r[idx])
    print(f"strip {s:3d} rows {idx[0]:4d}-{idx[-1]:4d} start {r[idx[0]]:8.1f} end {r[idx[-1]]:8.1f} period mean {d.mean():6.2f} first50 {d[:50].mean():6.2f} mid {d[len(d)//2-25:len(d)//2+25].mean():6.2f} last50 {d[-50:].mean():6.2f} max {d.max():7.1f}")
# lag between neighbours at same row
for i in [10,100,500,1000,2000]:
    col=st[:,i]; ok=~np.isnan(col)
    idx=np.where(ok)[0]
    if len(idx)>2:
        d=np.diff(col[idx]); print("row",i,"strips",idx[0],idx[-1],"lag per strip mean",np.nanmean(d),"median",np.nanmedian(d),"max",np.nanmax(d))
# detail: one strip in the middle of the run
for s in [150, 220]:
    if s >= ns: continue
    r = st[s]; l = st[s-1]
    rows = np.arange(1000, 1030)
    print("strip", s, "periods", np.round(np.diff(r[999:1030]), 1).tolist())
    print("strip", s, "lag to left", np.round((r - l)[rows], 1).tolist())
    d = np.diff(r[200:2000]); print("strip", s, "period percentiles 10/50/90/99", np.round(np.percentile(d, [10, 50, 90, 99]), 2).tolist())
# how many strips are between their first and last barrier at a given time
ends = np.nanmax(st, axis=1); starts = np.nanmin(np.where(st > 0, st, np.nan), axis=1)
for t in [500, 2000, 5000, 8000, 11000, 14000]:
    print("t", t, "active strips", int(np.sum((starts <= t) & (ends >= t))))
