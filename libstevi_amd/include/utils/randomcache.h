// Drop-in for LibStevi's utils/randomcache.h: StereoVision::Random::NumbersCache<T>, the table of pre-drawn numbers the reference's callers
// hand to patchMatch / cachelessPatchMatch as `randcache` (benchmarkStereoMatchingModels.cpp:176, :187-199).  Same interface (default
// constructor, (n, generator) constructor, size(), seed(idx), operator()); written against those call sites.  The GPU entry points accept
// a cache and do not draw from it -- the reference's draws depend on its OpenMP schedule, so its stream cannot be reproduced; see
// correlation/patchmatch.h -- but code that fills and passes one compiles and runs unchanged.
#ifndef STEREOVISION_RANDOMCACHE_H
#define STEREOVISION_RANDOMCACHE_H

#include <cstddef>
#include <functional>
#include <memory>
#include <vector>

namespace StereoVision {
namespace Random {

template <typename T> class NumbersCache {
  public:
    NumbersCache() : _values(std::make_shared<std::vector<T>>(1)), _next(0) {}
    NumbersCache(std::size_t n, std::function<T()> const &generator) : _values(std::make_shared<std::vector<T>>(n ? n : 1)), _next(0) {
        for (T &v : *_values) v = generator();
    }
    std::size_t size() const { return _values->size(); }
    // moves the read position by an amount that depends on idx and on the table's contents (callers use it to decorrelate threads)
    void seed(int idx) {
        std::size_t pos = _next + static_cast<std::size_t>(idx);
        for (int round = 0; round < 8; round++) pos += static_cast<std::size_t>((*_values)[pos % size()]) * 2654435761u + 1;
        _next = pos % size();
    }
    T operator()() const {
        _next = (_next + 1) % size();
        return (*_values)[_next];
    }

  private:
    std::shared_ptr<std::vector<T>> _values; // shared between copies, like the reference's table
    mutable std::size_t _next;
};

} // namespace Random
} // namespace StereoVision

#endif // STEREOVISION_RANDOMCACHE_H
