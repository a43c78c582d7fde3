// Does hipStreamWaitValue32 work on this box?  A kernel on stream b bumps a counter in signal memory; stream a waits for it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void bump(unsigned *c) { if (threadIdx.x == 0) atomicAdd(c, 1u); }
__global__ void mark(unsigned *c, unsigned *out) { if (threadIdx.x == 0) *out = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define P(x) do { hipError_t e_ = (x); printf("%s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); } while (0)
int main()
{
    int v = 0;
    P(hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue %d\n", v); fflush(stdout);
    unsigned *sig = nullptr, *out = nullptr;
    P(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    P(hipMalloc((void **)&out, 8));
    P(hipMemset(sig, 0, 8));
    P(hipMemset(out, 0, 8));
    hipStream_t a, b;
    P(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    P(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    P(hipStreamWaitValue32(a, sig, 64, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, a, sig, out);
    hipLaunchKernelGGL(bump, dim3(64), dim3(64), 0, b, sig);
    P(hipStreamSynchronize(b));
    P(hipStreamSynchronize(a));
    unsigned h = 0;
    P(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("marker saw %u (expected 64)\n", h);
    // plain device memory instead of signal memory
    unsigned *plain = nullptr;
    P(hipMalloc((void **)&plain, 8));
    P(hipMemset(plain, 0, 8));
    P(hipStreamWaitValue32(a, plain, 64, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, a, plain, out);
    hipLaunchKernelGGL(bump, dim3(64), dim3(64), 0, b, plain);
    P(hipStreamSynchronize(b));
    P(hipStreamSynchronize(a));
    P(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("plain memory: marker saw %u (expected 64)\n", h);
    return 0;
}
