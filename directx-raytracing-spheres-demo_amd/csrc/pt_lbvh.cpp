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

// ---- SAH builder --------------------------------------------------------------------------------------------------
// The topology for small scenes (DESIGN.md "LBVH": the analogue of the reference's PREFER_FAST_TRACE builds,
// Source/Scene.ixx:247,283).  Top-down surface-area heuristic over the spheres' boxes, one sphere per leaf: ranges of up to
// kSweepMax spheres are split by an exact sweep along each axis, larger ones by 32 bins per axis; below kBalanceLevel levels
// the split is the median of the widest axis, which bounds the depth (the traversal stack and the refit passes are sized by
// it).  Same record format as the LBVH -- nodes in pre-order, spheres in leaf order -- so every kernel runs unchanged; the
// device fills in the boxes (lbvh_gpu_adopt), the host boxes below are the same arithmetic.
namespace {

struct Box3 { float lo[3], hi[3]; };
inline void box_reset(Box3& b) { for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } }
inline void box_grow(Box3& b, const Box3& o) { for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); } }
inline float box_area(const Box3& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx < 0 ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
}

constexpr uint32_t kSweepMax = 32;  // measured on C2: sweeps up to 256 / 32 / 4 spheres render alike (0.1023 / 0.1033 / 0.1026 ms), 32 builds fastest
constexpr int kBins = 32;
constexpr uint32_t kBalanceLevel = 24;

struct SahBuilder {
    const PtSphere* sph;
    std::vector<Box3> boxes;
    std::vector<uint32_t> idx;
    std::vector<float> right_area;  // scratch of the sweep

    float centre(uint32_t id, int a) const { return a == 0 ? sph[id].cx : a == 1 ? sph[id].cy : sph[id].cz; }
    // total order along an axis (ties by id: the build must not depend on the sort implementation)
    void sort_axis(uint32_t lo, uint32_t hi, int a)
    {
        std::sort(idx.begin() + lo, idx.begin() + hi, [&](uint32_t x, uint32_t y) {
            const float cx = centre(x, a), cy = centre(y, a);
            return cx < cy || (cx == cy && x < y);
        });
    }

    // split [lo, hi) (>= 3 spheres) and return the first index of the right part; lo < result < hi
    uint32_t split(uint32_t lo, uint32_t hi, uint32_t level)
    {
        const uint32_t cnt = hi - lo;
        Box3 cb; box_reset(cb);
        for (uint32_t k = lo; k < hi; k++)
            for (int a = 0; a < 3; a++) { const float c = centre(idx[k], a); cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c); }
        int wide = 0;
        for (int a = 1; a < 3; a++) if (cb.hi[a] - cb.lo[a] > cb.hi[wide] - cb.lo[wide]) wide = a;
        if (level >= kBalanceLevel || !(cb.hi[wide] - cb.lo[wide] > 0)) {  // depth guard / coincident centres: median by count
            sort_axis(lo, hi, wide);
            return lo + cnt / 2;
        }
        float best = INFINITY; int best_axis = -1; uint32_t best_pos = 0;
        if (cnt <= kSweepMax) {
            right_area.resize(cnt);
            int sorted_by = -1;
            for (int a = 0; a < 3; a++) {
                if (!(cb.hi[a] - cb.lo[a] > 0)) continue;
                sort_axis(lo, hi, a);
                sorted_by = a;
                Box3 acc; box_reset(acc);
                for (uint32_t k = cnt - 1; k > 0; k--) { box_grow(acc, boxes[idx[lo + k]]); right_area[k] = box_area(acc); }
                box_reset(acc);
                for (uint32_t k = 1; k < cnt; k++) {  // left = [0, k), right = [k, cnt)
                    box_grow(acc, boxes[idx[lo + k - 1]]);
                    const float cost = box_area(acc) * (float)k + right_area[k] * (float)(cnt - k);
                    if (cost < best) { best = cost; best_axis = a; best_pos = k; }
                }
            }
            if (best_axis < 0) { sort_axis(lo, hi, wide); return lo + cnt / 2; }
            if (best_axis != sorted_by) sort_axis(lo, hi, best_axis);
            return lo + best_pos;
        }
        int best_bin = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Box3 bb[kBins]; uint32_t bc[kBins] = {};
            for (auto& b : bb) box_reset(b);
            const float scale = kBins / ext;
            for (uint32_t k = lo; k < hi; k++) {
                int b = (int)((centre(idx[k], a) - cb.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bc[b]++; box_grow(bb[b], boxes[idx[k]]);
            }
            float r_area[kBins]; uint32_t r_cnt[kBins];
            Box3 acc; box_reset(acc); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; b--) { box_grow(acc, bb[b]); c += bc[b]; r_area[b] = box_area(acc); r_cnt[b] = c; }
            box_reset(acc); c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                box_grow(acc, bb[b]); c += bc[b];
                if (c == 0 || r_cnt[b + 1] == 0) continue;
                const float cost = box_area(acc) * (float)c + r_area[b + 1] * (float)r_cnt[b + 1];
                if (cost < best) { best = cost; best_axis = a; best_bin = b; }
            }
        }
        if (best_axis >= 0) {
            const float scale = kBins / (cb.hi[best_axis] - cb.lo[best_axis]), origin = cb.lo[best_axis];
            const auto it = std::stable_partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t id) {
                int b = (int)((centre(id, best_axis) - origin) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            const uint32_t mid = (uint32_t)(it - idx.begin());
            if (mid > lo && mid < hi) return mid;
        }
        sort_axis(lo, hi, wide);
        return lo + cnt / 2;
    }
};

}  // namespace

void build_sah_host(const PtSphere* sph, uint32_t n, LbvhResult& out)
{
    out = LbvhResult{};
    if (n == 0) return;
    SahBuilder sb;
    sb.sph = sph;
    sb.boxes.resize(n);
    for (int a = 0; a < 3; a++) { out.bounds_min[a] = INFINITY; out.bounds_max[a] = -INFINITY; }
    for (uint32_t i = 0; i < n; i++) {
        const float c[3] = { sph[i].cx, sph[i].cy, sph[i].cz };
        for (int a = 0; a < 3; a++) {
            sb.boxes[i].lo[a] = c[a] - sph[i].r; sb.boxes[i].hi[a] = c[a] + sph[i].r;
            out.bounds_min[a] = std::min(out.bounds_min[a], sb.boxes[i].lo[a]);
            out.bounds_max[a] = std::max(out.bounds_max[a], sb.boxes[i].hi[a]);
        }
    }
    out.pad = lbvh_padding(out.bounds_min, out.bounds_max);
    sb.idx.resize(n);
    std::iota(sb.idx.begin(), sb.idx.end(), 0u);
    out.sorted.resize(n); out.sorted_id.resize(n);
    if (n == 1) { out.sorted[0] = sph[0]; out.sorted_id[0] = 0; return; }
    out.nodes.resize(n - 1);
    struct Pending { uint32_t lo, hi; int parent, slot; uint32_t level; };
    std::vector<Pending> todo{ { 0, n, -1, 0, 1 } };
    uint32_t next_node = 0, depth = 0;
    while (!todo.empty()) {
        const Pending t = todo.back();
        todo.pop_back();
        const uint32_t cnt = t.hi - t.lo;
        if (cnt == 1) {  // leaf: slot t.lo of the leaf order
            const uint32_t id = sb.idx[t.lo];
            out.sorted[t.lo] = sph[id]; out.sorted_id[t.lo] = id;
            (t.slot == 0 ? out.nodes[t.parent].child0 : out.nodes[t.parent].child1) = ~(int)t.lo;
            continue;
        }
        const int me = (int)next_node++;  // pre-order: a node precedes its subtrees, the left subtree precedes the right one
        depth = std::max(depth, t.level);
        out.nodes[me].parent = t.parent;
        out.nodes[me]._pad = 0;
        if (t.parent >= 0) (t.slot == 0 ? out.nodes[t.parent].child0 : out.nodes[t.parent].child1) = me;
        const uint32_t mid = cnt == 2 ? t.lo + 1 : sb.split(t.lo, t.hi, t.level);
        todo.push_back({ mid, t.hi, me, 1, t.level + 1 });
        todo.push_back({ t.lo, mid, me, 0, t.level + 1 });
    }
    out.depth = depth;
    // boxes bottom-up: in pre-order the children of a node have larger indices
    std::vector<Box3> nb(n - 1);
    for (int i = (int)n - 2; i >= 0; i--) {
        PtBvhNode& nd = out.nodes[i];
        auto child_box = [&](int c, float lo[3], float hi[3]) {
            if (c < 0) {
                const PtSphere& s = out.sorted[~c];
                const float cc[3] = { s.cx, s.cy, s.cz };
                for (int a = 0; a < 3; a++) { lo[a] = cc[a] - s.r - out.pad; hi[a] = cc[a] + s.r + out.pad; }
            } else {
                for (int a = 0; a < 3; a++) { lo[a] = nb[c].lo[a]; hi[a] = nb[c].hi[a]; }
            }
        };
        child_box(nd.child0, nd.lo0, nd.hi0);
        child_box(nd.child1, nd.lo1, nd.hi1);
        for (int a = 0; a < 3; a++) { nb[i].lo[a] = std::min(nd.lo0[a], nd.lo1[a]); nb[i].hi[a] = std::max(nd.hi0[a], nd.hi1[a]); }
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

extern "C" PtStatus pt_sah_build_host(const PtSphere* spheres, uint32_t n, PtBvhNode* nodes, uint32_t* sorted_id, uint32_t* depth)
{
    if (!spheres || n == 0 || !sorted_id || (n > 1 && !nodes)) return PT_ERR_INVALID_ARG;
    pt::LbvhResult r;
    pt::build_sah_host(spheres, n, r);
    for (uint32_t i = 0; i + 1 < n; i++) nodes[i] = r.nodes[i];
    for (uint32_t i = 0; i < n; i++) sorted_id[i] = r.sorted_id[i];
    if (depth) *depth = r.depth;
    return PT_OK;
}
