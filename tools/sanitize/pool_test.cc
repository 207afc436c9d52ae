// Stress of the host-side helper pool (gr-clenabled_amd/csrc/runtime.hip: mi355_copy / mi355_parallel) under ThreadSanitizer and
// AddressSanitizer: several caller threads (GNU Radio runs one thread per block) copy large buffers and run split jobs at the
// same time; a busy pool must make the caller fall back to copying alone, never corrupt or race.  No GPU needed.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

void mi355_copy(void *dst, const void *src, size_t bytes);
void mi355_copy2(void *dst0, const void *src0, size_t bytes0, void *dst1, const void *src1, size_t bytes1);
bool mi355_parallel(void (*fn)(void *, int part, int parts), void *arg);

struct Job { std::vector<int> *v; std::atomic<int> calls{0}; };
static void fill(void *a, int part, int parts)
{
    Job &j = *(Job *)a;
    const size_t n = j.v->size(), per = (n + parts - 1) / parts, b = part * per, e = b + per < n ? b + per : n;
    for (size_t i = b; i < e; i++) (*j.v)[i] = (int)i * 3 + 1;
    j.calls++;
}

int main()
{
    std::atomic<int> bad{0}, pooled{0};
    auto worker = [&](int id) {
        const size_t bytes = (size_t)(3 + id) << 20;  // 3..6 MiB: above the pool threshold
        std::vector<unsigned char> a(bytes), b(bytes);
        for (int rep = 0; rep < 40; rep++) {
            for (size_t i = 0; i < bytes; i += 4093) a[i] = (unsigned char)(i + rep + id);
            mi355_copy(b.data(), a.data(), bytes);
            if (memcmp(a.data(), b.data(), bytes)) bad++;
            // two copies as one job (a staging slot's copy-out and copy-in), unaligned starts and lengths, streaming-store path included
            {
                const size_t o0 = 1 + (size_t)rep % 61, o1 = 3 + (size_t)id * 7, n0 = bytes / 2 - 97 - o0, n1 = bytes / 2 - 4099 - o1;
                std::vector<unsigned char> c(bytes, 0xEE);
                mi355_copy2(c.data() + o0, a.data() + 5, n0, c.data() + bytes / 2 + o1, a.data() + bytes / 2 + 11, n1);
                if (memcmp(c.data() + o0, a.data() + 5, n0) || memcmp(c.data() + bytes / 2 + o1, a.data() + bytes / 2 + 11, n1)) bad++;
                if (c[o0 - 1] != 0xEE || c[o0 + n0] != 0xEE || c[bytes / 2 + o1 - 1] != 0xEE || c[bytes / 2 + o1 + n1] != 0xEE) bad++;  // nothing outside the ranges
                mi355_copy2(nullptr, nullptr, 0, c.data(), a.data(), 1000);  // an empty first copy, a small second one
                if (memcmp(c.data(), a.data(), 1000)) bad++;
            }
            std::vector<int> v(100000 + 1000 * id, 0);
            Job j;
            j.v = &v;
            if (mi355_parallel(fill, &j)) pooled++;
            else fill(&j, 0, 1);
            for (size_t i = 0; i < v.size(); i++)
                if (v[i] != (int)i * 3 + 1) { bad++; break; }
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < 4; i++) th.emplace_back(worker, i);
    for (auto &t : th) t.join();
    printf("pool stress: %d mismatches, %d split jobs ran on the pool\n", bad.load(), pooled.load());
    return bad.load() ? 1 : 0;
}
