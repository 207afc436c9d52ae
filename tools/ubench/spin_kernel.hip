// A kernel that holds its workgroups' CUs for a given time (see tools/xe_contention_probe.py): built with hipcc --genco into spin_kernel.co
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void k_spin(unsigned long long ticks, int *out)
{
    // (64 registers per lane: a CU that runs this workgroup has no room for a second, register-heavy one -- like a real neighbour)
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    const unsigned long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) n++;
    if (n == 0x7fffffff) out[0] = n;
}
