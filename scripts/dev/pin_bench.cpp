// how long does page-locking ~900 MB take, by method (hipHostMalloc, hipHostRegister of plain / huge-page-advised memory)?
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    void* d; hipMalloc(&d, 1 << 20);
    const size_t bytes = 900ull << 20;
    for (int rep = 0; rep < 2; ++rep) {
        double t = now();
        void* p = nullptr;
        hipHostMalloc(&p, bytes, hipHostMallocDefault);
        printf("hipHostMalloc                     %.1f ms\n", now() - t);
        t = now(); hipHostFree(p); printf("  free %.1f ms\n", now() - t);
        for (int huge = 0; huge < 2; ++huge) {
            t = now();
            void* q = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            char* a = (char*)(((uintptr_t)q + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
            if (huge) madvise(a, bytes, MADV_HUGEPAGE);
            double t1 = now();
            for (size_t i = 0; i < bytes; i += 4096) a[i] = 1;
            double t2 = now();
            hipError_t e = hipHostRegister(a, bytes, hipHostRegisterDefault);
            double t3 = now();
            printf("mmap%s: map %.1f touch %.1f register %.1f ms (%s)\n", huge ? "+MADV_HUGEPAGE" : "", t1 - t, t2 - t1, t3 - t2, hipGetErrorString(e));
            // D2H bandwidth into it
            void* dev; hipMalloc(&dev, 256 << 20);
            hipMemcpy(a, dev, 256 << 20, hipMemcpyDeviceToHost);
            t = now(); hipMemcpy(a, dev, 256 << 20, hipMemcpyDeviceToHost); printf("  D2H 256 MB %.1f ms\n", now() - t);
            hipFree(dev);
            t = now(); hipHostUnregister(a); munmap(q, bytes + (2 << 20)); printf("  unregister+unmap %.1f ms\n", now() - t);
        }
    }
    return 0;
}
