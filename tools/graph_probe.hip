// graph_probe.hip — what a hipGraph would buy the dispatch loop: host time to enqueue one dispatch's chain (three dependent
// kernels and one device-to-host copy on the frame's stream) as four stream operations against one hipGraphLaunch of the same
// chain captured once per frame, with the kernel parameters of the first node updated before every launch as a real dispatch
// would need (hipGraphExecKernelNodeSetParams: the chain's arguments change with every start key).  Stand-in kernels of the
// real durations (spin for 14 / 28 / 110 us on one wave per SIMD), twelve frames in flight, like the scan.
// Build: hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o tools/graph_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

struct Args {           // ~2.5 KB by value, like SeqArgs
    unsigned long long ticks;
    unsigned *out;
    unsigned pad[600];
};

__global__ void __launch_bounds__(256) spin(const Args a) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < a.ticks) {}
    if (a.pad[threadIdx.x % 600] == 0xFFFFFFFFu) a.out[0] = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int F = 12, STEPS = 600, WGS = 256;
    std::vector<hipStream_t> st(F);
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int order[3] = {0, -1, 1};
    for (int f = 0; f < F; f++) CHECK(hipStreamCreateWithPriority(&st[f], hipStreamNonBlocking, order[(f / 4) % 3]));
    unsigned *d = nullptr, *h = nullptr;
    CHECK(hipMalloc(&d, F * 16384));
    CHECK(hipHostMalloc(&h, F * 16384));
    Args a[3];
    for (int k = 0; k < 3; k++) {
        a[k].out = d;
        for (unsigned &p : a[k].pad) p = 0;
    }
    a[0].ticks = 1400;   // 100 MHz ticks: 14 us
    a[1].ticks = 2800;
    a[2].ticks = 11000;

    auto enqueue_plain = [&](int f) {
        hipLaunchKernelGGL(spin, dim3(WGS), dim3(256), 0, st[f], a[0]);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[f], a[1]);
        hipLaunchKernelGGL(spin, dim3(WGS), dim3(256), 0, st[f], a[2]);
        CHECK(hipMemcpyAsync(h + f * 4096, d + f * 4096, 10240, hipMemcpyDeviceToHost, st[f]));
    };
    // one graph per frame, captured from the same four operations
    std::vector<hipGraph_t> g(F);
    std::vector<hipGraphExec_t> ge(F);
    std::vector<hipGraphNode_t> first(F);
    for (int f = 0; f < F; f++) {
        CHECK(hipStreamBeginCapture(st[f], hipStreamCaptureModeThreadLocal));
        enqueue_plain(f);
        CHECK(hipStreamEndCapture(st[f], &g[f]));
        CHECK(hipGraphInstantiate(&ge[f], g[f], nullptr, nullptr, 0));
        size_t n = 0;
        CHECK(hipGraphGetNodes(g[f], nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n);
        CHECK(hipGraphGetNodes(g[f], nodes.data(), &n));
        first[f] = nullptr;
        for (auto nd : nodes) {
            hipGraphNodeType t;
            CHECK(hipGraphNodeGetType(nd, &t));
            if (t == hipGraphNodeTypeKernel && !first[f]) first[f] = nd;
        }
    }
    for (int mode = 0; mode < 3; mode++) {   // 0 plain launches, 1 graph launch, 2 graph launch + parameter update of one node
        for (int f = 0; f < F; f++) enqueue_plain(f);
        CHECK(hipDeviceSynchronize());
        double host = 0;
        const double t0 = now();
        for (int s = 0; s < STEPS; s++) {
            const int f = s % F;
            if (s >= F) CHECK(hipStreamSynchronize(st[f]));
            const double h0 = now();
            if (mode == 0) {
                enqueue_plain(f);
            } else {
                if (mode == 2) {
                    hipKernelNodeParams kp;
                    CHECK(hipGraphKernelNodeGetParams(first[f], &kp));
                    void *params[1] = {&a[0]};
                    kp.kernelParams = params;
                    CHECK(hipGraphExecKernelNodeSetParams(ge[f], first[f], &kp));
                }
                CHECK(hipGraphLaunch(ge[f], st[f]));
            }
            host += now() - h0;
        }
        CHECK(hipDeviceSynchronize());
        const double dt = now() - t0;
        printf("%-58s host %.1f us per dispatch; %d dispatches in %.1f ms = %.1f us per dispatch end to end\n",
               mode == 0 ? "four stream operations (3 launches + 1 copy)" : mode == 1 ? "one hipGraphLaunch" : "hipGraphExecKernelNodeSetParams + hipGraphLaunch",
               host / STEPS * 1e6, STEPS, dt * 1e3, dt / STEPS * 1e6);
    }
    return 0;
}
