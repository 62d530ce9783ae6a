// pt_lbvh.cpp -- host LBVH builder (see pt_lbvh.h).
#include "pt_lbvh.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace pt {

static inline uint32_t expand10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

uint32_t morton30(float x, float y, float z)
{
    auto q = [](float v) {
        float s = v * 1024.0f;
        s = s < 0.0f ? 0.0f : (s > 1023.0f ? 1023.0f : s);
        return (uint32_t)s;
    };
    return (expand10(q(x)) << 2) | (expand10(q(y)) << 1) | expand10(q(z));
}

float lbvh_padding(const float bmin[3], const float bmax[3])
{
    float s = 0.0f;
    for (int a = 0; a < 3; a++) s = std::max(s, std::max(std::fabs(bmin[a]), std::fabs(bmax[a])));
    return s * 7.62939453125e-06f;  // 2^-17
}

// number of leading bits two 64-bit keys share; -1 when j is out of range
static inline int delta(const std::vector<uint64_t>& keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(keys[i] ^ keys[j]);  // keys are unique, so the xor is non-zero
}

void build_lbvh_host(const PtSphere* sph, uint32_t n, LbvhResult& out)
{
    out = LbvhResult{};
    if (n == 0) return;
    // scene bounds (with radii) and centroid bounds
    float cmin[3] = { INFINITY, INFINITY, INFINITY }, cmax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int a = 0; a < 3; a++) { out.bounds_min[a] = INFINITY; out.bounds_max[a] = -INFINITY; }
    for (uint32_t i = 0; i < n; i++) {
        const float c[3] = { sph[i].cx, sph[i].cy, sph[i].cz };
        for (int a = 0; a < 3; a++) {
            cmin[a] = std::min(cmin[a], c[a]); cmax[a] = std::max(cmax[a], c[a]);
            out.bounds_min[a] = std::min(out.bounds_min[a], c[a] - sph[i].r);
            out.bounds_max[a] = std::max(out.bounds_max[a], c[a] + sph[i].r);
        }
    }
    out.pad = lbvh_padding(out.bounds_min, out.bounds_max);

    // Morton keys: (code << 32) | original index  -> unique, sorted ascending
    std::vector<uint64_t> keys(n);
    float inv[3];
    for (int a = 0; a < 3; a++) { const float e = cmax[a] - cmin[a]; inv[a] = e > 0.0f ? 1.0f / e : 0.0f; }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t code = morton30((sph[i].cx - cmin[0]) * inv[0], (sph[i].cy - cmin[1]) * inv[1], (sph[i].cz - cmin[2]) * inv[2]);
        keys[i] = ((uint64_t)code << 32) | i;
    }
    std::sort(keys.begin(), keys.end());
    out.sorted.resize(n);
    out.sorted_id.resize(n);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t id = (uint32_t)(keys[k] & 0xFFFFFFFFu);
        out.sorted_id[k] = id;
        out.sorted[k] = sph[id];
    }
    if (n == 1) { out.depth = 0; return; }

    // Karras 2012: internal node i covers a key range determined by the common-prefix function delta
    const int N = (int)n;
    out.nodes.resize(n - 1);
    for (int i = 0; i < N - 1; i++) {
        const int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1) > 0 ? 1 : -1;
        const int dmin = delta(keys, N, i, i - d);
        int lmax = 2;
        while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(keys, N, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + std::min(d, 0);
        const int lo = std::min(i, j), hi = std::max(i, j);
        PtBvhNode& nd = out.nodes[i];
        nd.child0 = (lo == gamma) ? ~gamma : gamma;              // leaf k encoded as ~k (sorted index)
        nd.child1 = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
        nd._pad = 0;
    }
    out.nodes[0].parent = -1;
    for (int i = 0; i < N - 1; i++) {
        if (out.nodes[i].child0 >= 0) out.nodes[out.nodes[i].child0].parent = i;
        if (out.nodes[i].child1 >= 0) out.nodes[out.nodes[i].child1].parent = i;
    }

    // bottom-up AABBs: iterative post-order from the root
    auto leaf_box = [&](int k, float lo[3], float hi[3]) {
        const PtSphere& s = out.sorted[k];
        const float c[3] = { s.cx, s.cy, s.cz };
        for (int a = 0; a < 3; a++) { lo[a] = c[a] - s.r - out.pad; hi[a] = c[a] + s.r + out.pad; }
    };
    std::vector<int> order;  // internal nodes in pre-order; reversed = children before parents
    order.reserve(n - 1);
    std::vector<int> stack{ 0 };
    std::vector<uint32_t> level(n - 1, 0);
    level[0] = 1;
    uint32_t depth = 1;
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        order.push_back(i);
        for (int c : { out.nodes[i].child0, out.nodes[i].child1 })
            if (c >= 0) { level[c] = level[i] + 1; depth = std::max(depth, level[c]); stack.push_back(c); }
    }
    out.depth = depth;
    std::vector<float> nlo((size_t)(n - 1) * 3), nhi((size_t)(n - 1) * 3);
    for (auto it = order.rbegin(); it != order.rend(); ++it) {
        const int i = *it;
        PtBvhNode& nd = out.nodes[i];
        auto child_box = [&](int c, float lo[3], float hi[3]) {
            if (c < 0) leaf_box(~c, lo, hi);
            else for (int a = 0; a < 3; a++) { lo[a] = nlo[(size_t)c * 3 + a]; hi[a] = nhi[(size_t)c * 3 + a]; }
        };
        child_box(nd.child0, nd.lo0, nd.hi0);
        child_box(nd.child1, nd.lo1, nd.hi1);
        for (int a = 0; a < 3; a++) {
            nlo[(size_t)i * 3 + a] = std::min(nd.lo0[a], nd.lo1[a]);
            nhi[(size_t)i * 3 + a] = std::max(nd.hi0[a], nd.hi1[a]);
        }
    }
}

}  // namespace pt

extern "C" PtStatus pt_lbvh_build_host(const PtSphere* spheres, uint32_t n, PtBvhNode* nodes, uint32_t* sorted_id, uint32_t* depth)
{
    if (!spheres || n == 0 || !sorted_id || (n > 1 && !nodes)) return PT_ERR_INVALID_ARG;
    pt::LbvhResult r;
    pt::build_lbvh_host(spheres, n, r);
    for (uint32_t i = 0; i + 1 < n; i++) nodes[i] = r.nodes[i];
    for (uint32_t i = 0; i < n; i++) sorted_id[i] = r.sorted_id[i];
    if (depth) *depth = r.depth;
    return PT_OK;
}
