// Stress of the host-side helper pool (gr-clenabled_amd/csrc/runtime.hip: mi355_copy / mi355_parallel) under ThreadSanitizer and
// AddressSanitizer: several caller threads (GNU Radio runs one thread per block) copy large buffers and run split jobs at the
// same time; a busy pool must make the caller fall back to copying alone, never corrupt or race.  No GPU needed.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

void mi355_copy(void *dst, const void *src, size_t bytes);
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
