// chain_sort.hip — value index for the chaining traceback: for one tree kind of one chain combination, the records'
// stored values (order-preserving integer encoding) sorted together with their record numbers (cl_radix.h).
// The host then finds "all predecessors whose stored value equals this query's maximum" by binary search instead of
// scanning a million records per traceback step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cl_radix.h"

namespace {
// keys go through the sort as UNSIGNED words (the order-preserving signed encoding with its sign bit flipped) and come out as the signed encoding again
__global__ void chain_keys_kernel(const float* __restrict__ val, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    int b = __float_as_int(val[r]);
    if (b == (int)0x80000000) b = 0;   // -0.0f == +0.0f, as in the DP kernels' encoding
    keys[r] = (uint32_t)(b >= 0 ? b : b ^ 0x7FFFFFFF) ^ 0x80000000u;
    idx[r] = r;
}
__global__ void chain_keys_back_kernel(uint32_t* __restrict__ keys, uint32_t n) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) keys[r] ^= 0x80000000u;
}
}  // namespace

// sorts (enc(val[r]), r) by key; temp storage is (re)allocated by the caller through the two-call protocol
hipError_t cl_chain_sort_values(const float* val, uint32_t n, int* keys_in, uint32_t* idx_in, int* keys_out, uint32_t* idx_out,
                                void* temp, size_t* temp_bytes, hipStream_t stream) {
    if (temp == nullptr) { *temp_bytes = clradix::sort_temp_bytes<uint32_t>(n); return hipSuccess; }
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(chain_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, val, n, reinterpret_cast<uint32_t*>(keys_in), idx_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = clradix::sort_pairs<uint32_t>(temp, *temp_bytes, reinterpret_cast<const uint32_t*>(keys_in), reinterpret_cast<uint32_t*>(keys_out), idx_in, idx_out, (size_t)n, 0u, 32u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(chain_keys_back_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(keys_out), n);
    return hipGetLastError();
}
