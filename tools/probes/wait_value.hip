// Probe: can a stream be held back by hipStreamWaitValue32 until a kernel on another stream writes a flag?
// (candidate mechanism: start launch k+1 when launch k's work queue has run dry)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(unsigned* flag, unsigned* log, long long spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    log[0] = 1;                                              // "producer reached its signal point"
    __hip_atomic_store(flag, 7u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t1 = wall_clock64();
    while (wall_clock64() - t1 < spin) { }                   // keeps running after the signal (the "tail")
    log[1] = 1;
}
__global__ void consumer(const unsigned* log, unsigned* out) { out[0] = 100u + log[0] * 10u + log[1]; }   // 110: started after the signal, before the producer's end

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* flag = nullptr; unsigned *log = nullptr, *out = nullptr;
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMalloc((void**)&log, 8)); CK(hipMalloc((void**)&out, 4));
    CK(hipMemset(flag, 0, 8)); CK(hipMemset(log, 0, 8)); CK(hipMemset(out, 0, 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamWaitValue32(s1, flag, 7u, hipStreamWaitValueGte, 0xFFFFFFFFu));
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, s1, log, out);
    CK(hipEventRecord(e1, s1));
    hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, s2, flag, log, 200000LL);   // 2 ms at 100 MHz before the signal, 2 ms after
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    unsigned h = 0; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("consumer saw %u (110 = ran between the signal and the producer's end; 111 = after the end; 100 = before the signal)\n", h);
    return 0;
}
