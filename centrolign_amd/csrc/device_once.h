// device_once.h — run a piece of per-DEVICE setup the first time a thread launches on that device.
// hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) applies to the current device only, so the ">64 KB of dynamic LDS"
// opt-ins of the large kernels cannot be once per process (worker contexts of one process may sit on several GPUs: cl_msa_params.devices).
#ifndef CL_DEVICE_ONCE_H
#define CL_DEVICE_ONCE_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

struct ClDeviceOnce {
    std::atomic<uint64_t> done{0};   // bit d: the setup has run on device d (ordinals beyond 63 repeat it on every launch: harmless, idempotent)
    std::mutex m;
    template <class F>
    void operator()(F setup) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = dev < 64 ? 1ull << dev : 0;
        if (bit && (done.load(std::memory_order_acquire) & bit)) return;
        std::lock_guard<std::mutex> lock(m);
        if (bit && (done.load(std::memory_order_relaxed) & bit)) return;
        setup();
        done.fetch_or(bit, std::memory_order_release);
    }
};

#endif
