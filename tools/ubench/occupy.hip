// Holds K compute units busy for S seconds (one 1024-thread workgroup each, spinning on the wall clock): a stand-in for another stream's
// kernels (an RCCL exchange) next to a launch that wants every CU -- used to check the X-engine's in-launch reduction when not all of its
// workgroups are resident.   usage: occupy <K> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void k_spin(unsigned long long ticks, int *out)
{
    const unsigned long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) n++;
    if (n == 0x7fffffff) out[0] = n;
}
int main(int argc, char **argv)
{
    const int k = argc > 1 ? atoi(argv[1]) : 16;
    const double s = argc > 2 ? atof(argv[2]) : 2.0;
    int *out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    // in pieces of 20 ms so that the device stays responsive to the other process's launches
    for (double done = 0; done < s; done += 0.02) {
        hipLaunchKernelGGL(k_spin, dim3(k), dim3(1024), 0, 0, (unsigned long long)(0.02 * 1e8), out);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
    }
    printf("occupied %d CUs for %.1f s\n", k, s);
    return 0;
}
