// Stand-in for the two hipCUB calls lbvh.hip makes, for the CPU build of the library (tests/cpp/emu_rt.h): "device" memory is
// host memory there and every call is synchronous.  TEST INFRASTRUCTURE; found ahead of /opt/rocm/include by the emulation
// build only.
#ifndef PTC_EMU_HIPCUB_HPP
#define PTC_EMU_HIPCUB_HPP
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <vector>

namespace hipcub {
struct DeviceRadixSort {
    // stable ascending sort of (key, value) pairs on key bits [begin_bit, end_bit); temp == nullptr: size query
    template <class K, class V, class Stream>
    static hipError_t SortPairs(void* temp, size_t& temp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, int n,
                                int begin_bit, int end_bit, Stream) {
        if (!temp) { temp_bytes = 1; return hipSuccess; }
        const K mask = (end_bit - begin_bit >= (int)(8 * sizeof(K))) ? ~K(0) : (((K(1) << (end_bit - begin_bit)) - 1) << begin_bit);
        std::vector<uint32_t> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
        for (int i = 0; i < n; ++i) { keys_out[i] = keys_in[idx[(size_t)i]]; vals_out[i] = vals_in[idx[(size_t)i]]; }
        return hipSuccess;
    }
};
struct DeviceScan {
    template <class In, class Out, class Stream>
    static hipError_t ExclusiveSum(void* temp, size_t& temp_bytes, const In* in, Out* out, int n, Stream) {
        if (!temp) { temp_bytes = 1; return hipSuccess; }
        Out acc = 0;
        for (int i = 0; i < n; ++i) { const Out v = (Out)in[i]; out[i] = acc; acc += v; }
        return hipSuccess;
    }
};
}  // namespace hipcub
#endif
