// The mixed-radix planner of fft_mr.hip (host code: factorisation search, quotient constants, twiddle runs) under ASan + UBSan:
// every length 2 .. 20000, both variants and both directions; checks the invariants the kernel relies on.
#include <cstdio>
#include <vector>

#include "fft_mr.h"

int main()
{
    long plans = 0, bad = 0;
    for (int n = 2; n <= 20000; n++)
        for (int variant = 0; variant < 2; variant++)
            for (int sign = -1; sign <= 1; sign += 2) {
                MrPlan p;
                std::vector<float> tw;
                if (!mi355_fft_mr_plan(n, sign, variant, &p, &tw)) continue;
                plans++;
                long prod = 1, ns = 1;
                size_t entries = 0;
                for (int i = 0; i < p.npass; i++) {
                    const MrPass &ps = p.pass[i];
                    if (ps.ns != ns || ps.nb != n / ps.radix || (i > 0 && (size_t)ps.tw_off != entries)) bad++;
                    // the multiply-high quotients must be exact for every butterfly number a workgroup can see
                    for (unsigned b = 0; b < (unsigned)(p.threads * 16 / 2 + 64); b += 37) {
                        if ((unsigned)(((unsigned long long)b * ps.m_nb) >> 32) != b / (unsigned)ps.nb) bad++;
                        if (i > 0 && (unsigned)(((unsigned long long)b * ps.m_ns) >> 32) != b / (unsigned)ps.ns) bad++;
                    }
                    if (i > 0) entries += (size_t)ns;
                    prod *= ps.radix;
                    ns *= ps.radix;
                }
                if (prod != n || entries * 2 != tw.size() || p.threads % 64 || p.threads > 1024 || p.frames < 1 ||
                    (long)p.frames * n > (long)p.threads * p.per_thread || p.lds_bytes > 160 * 1024)
                    bad++;
            }
    printf("mixed-radix planner: %ld plans, %ld violations\n", plans, bad);
    return bad ? 1 : 0;
}
