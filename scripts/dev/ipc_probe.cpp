// probe: can two processes on ONE device share device memory (hipIpc*) and order their streams through memory
// (hipStreamWriteValue32 / hipStreamWaitValue32 on IPC-mapped words)?   usage: ipc_probe a|b <file>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s failed: %s\n", role, #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void fill(uint32_t* p, uint32_t n, uint32_t v) { uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void check(const uint32_t* p, uint32_t n, uint32_t v, uint32_t* bad) { uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n && p[i] != v + i) atomicAdd(bad, 1u); }
struct Handles { hipIpcMemHandle_t buf, flags; };
int main(int argc, char** argv) {
    const char* role = argv[1];
    const char* path = argv[2];
    const uint32_t n = 1 << 20;
    CK(hipSetDevice(0));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("%s: CanUseStreamWaitValue = %d\n", role, can);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (role[0] == 'a') {
        uint32_t *buf, *flags, *bad;
        CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&bad, 4));
        CK(hipMemset(buf, 0, n * 4)); CK(hipMemset(flags, 0, 4096)); CK(hipMemset(bad, 0, 4));
        CK(hipDeviceSynchronize());
        Handles h;
        CK(hipIpcGetMemHandle(&h.buf, buf)); CK(hipIpcGetMemHandle(&h.flags, flags));
        FILE* f = fopen(path, "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f);
        std::string done = std::string(path) + ".ready"; f = fopen(done.c_str(), "wb"); fclose(f);
        for (uint32_t round = 1; round <= 200; ++round) {
            // wait for b's write of this round, check it, answer through flags[1]
            CK(hipStreamWaitValue32(s, flags, round, hipStreamWaitValueGte));
            hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, s, buf, n, round * 7u, bad);
            CK(hipStreamWriteValue32(s, flags + 1, round, 0));
        }
        auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        uint32_t hbad = 99; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("a: 200 rounds done, mismatches %u (sync waited %.1f ms)\n", hbad, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        sleep(1);
        return hbad ? 1 : 0;
    } else {
        std::string done = std::string(path) + ".ready";
        for (int i = 0; i < 600 && access(done.c_str(), F_OK) != 0; ++i) usleep(100000);
        Handles h; FILE* f = fopen(path, "rb"); if (!f || fread(&h, sizeof(h), 1, f) != 1) { printf("b: no handles\n"); return 2; } fclose(f);
        uint32_t *buf, *flags;
        CK(hipIpcOpenMemHandle((void**)&buf, h.buf, hipIpcMemLazyEnablePeerAccess));
        CK(hipIpcOpenMemHandle((void**)&flags, h.flags, hipIpcMemLazyEnablePeerAccess));
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t round = 1; round <= 200; ++round) {
            if (round > 1) CK(hipStreamWaitValue32(s, flags + 1, round - 1, hipStreamWaitValueGte));   // a has checked the round before
            hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, s, buf, n, round * 7u);
            CK(hipStreamWriteValue32(s, flags, round, 0));
        }
        CK(hipStreamSynchronize(s));
        printf("b: 200 rounds enqueued and done in %.1f ms (%.1f us per round trip)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
        CK(hipIpcCloseMemHandle(buf)); CK(hipIpcCloseMemHandle(flags));
        return 0;
    }
}
