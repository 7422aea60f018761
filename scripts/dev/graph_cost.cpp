// What would a hipGraph of the chaining DP's macro-block loop cost?  ("measure a hipGraph per 64-128 macro-blocks before declining it again".)
// The loop of chain_dp_batch (cl_chain_api.cpp) per macro-block k: far(k) on a side stream behind seal(k - lag - 1); near(k), walk(k) on the serial stream, walk behind
// far(k); seal(k) on the sealing stream behind walk(k): four launches, three cross-stream edges.  This program builds exactly that dependency pattern out of empty
// kernels (plus a ~20 us spin kernel for "walk", so that the device side is a chain as in the DP) and times, for B macro-blocks:
//   (a) direct enqueue (hipExtLaunchKernel with stop events + hipStreamWaitEvent, as the library does): host time to enqueue, time until the device is done;
//   (b) the same calls under stream capture -> hipGraphInstantiate -> hipGraphLaunch: capture, instantiate, launch + wait;  and a second launch of the same executable graph.
// A DP's arguments change with every DP (pointers, counts), so (b) pays capture + instantiate per DP unless every node is patched (hipGraphExecKernelNodeSetParams: one
// call per node again).   build: hipcc -O2 --offload-arch=gfx950 scripts/dev/graph_cost.cpp -o /tmp/graph_cost ; run: /tmp/graph_cost [blocks]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
// the library's kernels take their descriptors BY VALUE (ClChainDevice + ClFarDevice: ~760 bytes of kernel arguments): does that cost host time per launch?  (third argument "big")
struct BigArgs { int x[190]; };
__global__ void empty_big_kernel(BigArgs a, int* p) { if (p && threadIdx.x == 9999) *p = a.x[7]; }
__global__ void spin_kernel(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }   // 100 MHz ticks

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, lag = 2;
    const bool big = argc > 2 && argv[2][0] == 'b';
    BigArgs ba{};
    hipStream_t s0, far[2], seal;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    for (auto& s : far) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&seal, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev_walk(B), ev_far(B), ev_seal(B);
    for (int k = 0; k < B; ++k) {
        CK(hipEventCreateWithFlags(&ev_walk[k], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_far[k], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_seal[k], hipEventDisableTiming));
    }
    hipEvent_t fork, join_far[2], join_seal;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto& e : join_far) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_seal, hipEventDisableTiming));
    auto enqueue = [&](bool ext) {
        CK(hipEventRecord(fork, s0));
        for (auto& s : far) CK(hipStreamWaitEvent(s, fork, 0));
        CK(hipStreamWaitEvent(seal, fork, 0));
        for (int k = 0; k < B; ++k) {
            hipStream_t fs = far[k & 1];
            if (k > lag) {
                CK(hipStreamWaitEvent(fs, ev_seal[k - lag - 1], 0));
                if (ext && big) hipExtLaunchKernelGGL(empty_big_kernel, dim3(64), dim3(256), 0, fs, nullptr, ev_far[k], 0, ba, (int*)nullptr);
                else if (ext) hipExtLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, fs, nullptr, ev_far[k], 0, (int*)nullptr);
                else { hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, fs, (int*)nullptr); CK(hipEventRecord(ev_far[k], fs)); }
            }
            if (big) hipLaunchKernelGGL(empty_big_kernel, dim3(64), dim3(256), 0, s0, ba, (int*)nullptr);
            else hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, s0, (int*)nullptr);          // near
            if (k > lag) CK(hipStreamWaitEvent(s0, ev_far[k], 0));
            if (ext) hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, nullptr, ev_walk[k], 0, 2000LL);   // walk: 20 us
            else { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, 2000LL); CK(hipEventRecord(ev_walk[k], s0)); }
            CK(hipStreamWaitEvent(seal, ev_walk[k], 0));
            if (ext && big) hipExtLaunchKernelGGL(empty_big_kernel, dim3(8), dim3(64), 0, seal, nullptr, ev_seal[k], 0, ba, (int*)nullptr);
            else if (ext) hipExtLaunchKernelGGL(empty_kernel, dim3(8), dim3(64), 0, seal, nullptr, ev_seal[k], 0, (int*)nullptr);
            else { hipLaunchKernelGGL(empty_kernel, dim3(8), dim3(64), 0, seal, (int*)nullptr); CK(hipEventRecord(ev_seal[k], seal)); }
        }
        for (int f = 0; f < 2; ++f) { CK(hipEventRecord(join_far[f], far[f])); CK(hipStreamWaitEvent(s0, join_far[f], 0)); }
        CK(hipEventRecord(join_seal, seal)); CK(hipStreamWaitEvent(s0, join_seal, 0));
    };
    // warm-up
    enqueue(true); CK(hipStreamSynchronize(s0));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now_ms();
        enqueue(true);
        double t1 = now_ms();
        CK(hipStreamSynchronize(s0));
        double t2 = now_ms();
        printf("direct%s, %d macro-blocks: enqueue %.2f ms (%.1f us per block), until done %.2f ms\n", big ? " (760-byte arguments)" : "", B, t1 - t0, (t1 - t0) * 1e3 / B, t2 - t0);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipGraph_t g; hipGraphExec_t ge;
        double t0 = now_ms();
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        enqueue(false);   // (stop events of hipExtLaunchKernel are not captured: plain launches + hipEventRecord)
        CK(hipStreamEndCapture(s0, &g));
        double t1 = now_ms();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double t2 = now_ms();
        CK(hipGraphLaunch(ge, s0));
        double t3 = now_ms();
        CK(hipStreamSynchronize(s0));
        double t4 = now_ms();
        CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        double t5 = now_ms();
        size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
        printf("graph, %d macro-blocks, %zu nodes: capture %.2f ms, instantiate %.2f ms, launch call %.2f ms, until done %.2f ms (total %.2f ms = %.1f us per block); second launch of the same graph %.2f ms\n",
               B, nn, t1 - t0, t2 - t1, t3 - t2, t4 - t2, t4 - t0, (t4 - t0) * 1e3 / B, t5 - t4);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
