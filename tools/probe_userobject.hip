// probe_userobject.hip -- does a hipGraphExec keep the user objects of the graph it was instantiated from alive after
// hipGraphDestroy(graph)?  (CUDA: yes -- the executable holds references of its own.)  The answer decides whether the library
// may hand a capture chain's scratch buffer back when the GRAPH dies.
//   hipcc --offload-arch=gfx950 tools/probe_userobject.hip -o tools/probe_userobject && tools/probe_userobject
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static std::atomic<int> g_dead{0};
static void on_dead(void*) { g_dead.store(1); }
__global__ void k(int* p) { atomicAdd(p, 1); }
static bool dead_after(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); return g_dead.load() != 0; }
int main()
{
    int* d = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&d), 4));
    CK(hipMemset(d, 0, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d);
    hipStreamCaptureStatus st;
    unsigned long long id = 0;
    hipGraph_t cg = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &cg, &deps, &ndeps));
    std::printf("capturing graph handle during capture: %p\n", static_cast<void*>(cg));
    hipUserObject_t obj;
    CK(hipUserObjectCreate(&obj, nullptr, on_dead, 1, hipUserObjectNoDestructorSync));
    CK(hipGraphRetainUserObject(cg, obj, 1, hipGraphUserObjectMove));
    hipGraph_t g;
    CK(hipStreamEndCapture(s, &g));
    std::printf("graph from EndCapture == handle seen during capture: %d\n", g == cg);
    hipGraphExec_t ex;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    std::printf("after instantiate: destructor ran = %d\n", dead_after(50));
    CK(hipGraphDestroy(g));
    std::printf("after hipGraphDestroy(graph), executable alive: destructor ran = %d\n", dead_after(200));
    CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    std::printf("after a replay of the executable: destructor ran = %d\n", dead_after(50));
    CK(hipGraphExecDestroy(ex));
    std::printf("after hipGraphExecDestroy: destructor ran = %d\n", dead_after(200));
    int h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    std::printf("kernel ran %d time(s)\n", h);
    return 0;
}
