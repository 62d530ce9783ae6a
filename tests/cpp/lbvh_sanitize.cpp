// lbvh_sanitize.cpp -- the host BVH builders (csrc/pt_lbvh.cpp: Morton LBVH and SAH topology) under AddressSanitizer +
// UndefinedBehaviorSanitizer (tools/sanitize.sh): random, degenerate and clustered sphere sets, structural checks of the result.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../directx-raytracing-spheres-demo_amd/csrc/pt_lbvh.cpp"

static int Check(const pt::LbvhResult& r, uint32_t n, const char* what)
{
    if (r.sorted_id.size() != n || r.sorted.size() != n || r.nodes.size() != (n > 1 ? n - 1 : 0)) { std::fprintf(stderr, "%s n=%u: sizes\n", what, n); return 1; }
    std::vector<int> seen(n, 0), refs(n > 1 ? n - 1 : 0, 0);
    for (uint32_t id : r.sorted_id) { if (id >= n || seen[id]++) { std::fprintf(stderr, "%s n=%u: sorted_id is not a permutation\n", what, n); return 1; } }
    std::vector<int> leaf(n, 0);
    for (size_t i = 0; i < r.nodes.size(); i++)
        for (int c : { r.nodes[i].child0, r.nodes[i].child1 }) {
            if (c >= 0) { if ((size_t)c >= r.nodes.size() || refs[c]++) { std::fprintf(stderr, "%s n=%u: bad internal child\n", what, n); return 1; } }
            else { const uint32_t k = ~(uint32_t)c; if (k >= n || leaf[k]++) { std::fprintf(stderr, "%s n=%u: bad leaf\n", what, n); return 1; } }
        }
    for (uint32_t k = 0; k < n && n > 1; k++) if (!leaf[k]) { std::fprintf(stderr, "%s n=%u: leaf %u unreferenced\n", what, n, k); return 1; }
    return 0;
}

int main()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    int failed = 0, cases = 0;
    for (uint32_t n : { 1u, 2u, 3u, 7u, 64u, 441u, 1000u, 4096u, 20000u })
        for (int style = 0; style < 4; style++) {
            std::vector<PtSphere> s(n);
            for (auto& p : s) {
                const float spread = style == 1 ? 1e-3f : (style == 2 ? 1e3f : 10.0f);
                p.cx = u(rng) * spread; p.cy = u(rng) * spread; p.cz = style == 3 ? 0.0f : u(rng) * spread;  // style 3: coplanar, many equal Morton codes
                p.r = style == 3 ? 0.5f : 0.01f + 0.5f * (u(rng) + 1.0f);
            }
            if (style == 3) for (uint32_t i = 1; i < n; i += 2) s[i] = s[i - 1];  // exact duplicates
            pt::LbvhResult a, b;
            pt::build_lbvh_host(s.data(), n, a);
            failed += Check(a, n, "lbvh"); cases++;
            if (n <= 4096) { pt::build_sah_host(s.data(), n, b); failed += Check(b, n, "sah"); cases++; }
        }
    std::printf("%d builder cases, %d failed\n", cases, failed);
    return failed ? 1 : 0;
}
