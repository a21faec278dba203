// TEST: lsort::sort (moni_align_amd/csrc/sort_emul.h) must permute exactly like libstdc++ std::sort, ties included.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../moni_align_amd/csrc/sort_emul.h"

struct E { int key; int id; };

static std::vector<int> killer(int n) {   // median-of-3 killer (Musser): drives introsort into its heap-sort fallback
    std::vector<int> a(n);
    int k = n / 2;
    for (int i = 0; i < k; ++i) { if (i % 2 == 0) a[i] = i + 1; else a[i] = k + i + (k % 2 ? 0 : 1) ; }
    for (int i = 0; i < k; ++i) a[k + i] = 2 * (i + 1);
    return a;
}

int main() {
    std::mt19937 rng(12345);
    long checked = 0;
    for (int iter = 0; iter < 20000; ++iter) {
        int n = iter < 2000 ? (int)(rng() % 40) : (int)(rng() % 700);
        int mod = 1 + (int)(rng() % (iter % 3 == 0 ? 4 : (iter % 3 == 1 ? 50 : 100000)));
        std::vector<E> a(n), b;
        for (int i = 0; i < n; ++i) a[i] = E{(int)(rng() % mod), i};
        int kind = iter % 7;
        if (kind == 3) std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key < y.key; });
        if (kind == 4) std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key > y.key; });
        if (kind == 5 && n > 4) { auto k = killer(n); for (int i = 0; i < n; ++i) a[i].key = k[i]; }
        if (kind == 6 && n > 8) { for (int i = 0; i < n; ++i) a[i].key = (i < n / 2) ? i : n - i; }      // organ pipe
        for (int i = 0; i < n; ++i) a[i].id = i;
        b = a;
        bool desc = (iter & 1);
        if (desc) {
            std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key > y.key; });
            lsort::sort(b.data(), (long)n, [](const E& x, const E& y) { return x.key > y.key; });
        } else {
            std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key < y.key; });
            lsort::sort(b.data(), (long)n, [](const E& x, const E& y) { return x.key < y.key; });
        }
        for (int i = 0; i < n; ++i)
            if (a[i].key != b[i].key || a[i].id != b[i].id) { printf("MISMATCH iter %d n %d at %d\n", iter, n, i); return 1; }
        checked += n;
    }
    // large killer inputs (depth limit reached for sure)
    for (int n : {1000, 4096, 10007, 65536}) {
        auto k = killer(n);
        std::vector<E> a(n), b;
        for (int i = 0; i < n; ++i) a[i] = E{k[i] / 3, i};
        b = a;
        std::sort(a.begin(), a.end(), [](const E& x, const E& y) { return x.key < y.key; });
        lsort::sort(b.data(), (long)n, [](const E& x, const E& y) { return x.key < y.key; });
        for (int i = 0; i < n; ++i) if (a[i].id != b[i].id) { printf("MISMATCH killer n %d at %d\n", n, i); return 1; }
    }
    printf("OK %ld elements\n", checked);
    return 0;
}
