// tile_exchange_test.cpp -- the C++ host's multi-GPU frame exchange (host/TileExchange.hpp) without GPUs: `world` ranks live in
// one process, each with a host-memory backend that "renders" a known function of (x, y, frame) into the packed tile layout of
// pt_render_tiles, packs / un-swizzles exactly as the C-ABI documents (include/pt_api.h), and gathers through an in-process
// stand-in for pt_gather.  After every flushed batch rank 0's frames must hold that function for every pixel.
// Covers: root weights 0 / 1 / 2 / 5, 12-byte and 16-byte pixels, batches with a partial last one, ragged frame sizes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "TileExchange.hpp"

namespace {

constexpr uint32_t kTs = 32;

float Pattern(uint32_t x, uint32_t y, uint32_t frame, int ch) { return float((x * 131u + y * 7919u + frame * 104729u + uint32_t(ch) * 17u) % 65521u) * 0.25f + 1.0f; }

struct Network {  // what pt_gather does over RCCL, in one process: non-root ranks post their send buffer, the root copies
    std::vector<const void*> posted;
};

class FakeBackend {
public:
    FakeBackend(Network& net, uint32_t rank, uint32_t world, uint32_t w, uint32_t h) : m_net(net), m_rank(rank), m_world(world), m_w(w), m_h(h) {}
    void* Alloc(size_t bytes) { void* p = std::calloc(1, bytes ? bytes : 16); if (!p) throw std::bad_alloc(); return p; }
    void Free(void* p) noexcept { std::free(p); }
    void SetPartition(dxrs::tiles::Range r) { m_range = r; }
    void RenderTiles(uint32_t frame, void* out)
    {
        float* o = static_cast<float*>(out);
        const uint32_t tx = (m_w + kTs - 1) / kTs, total = dxrs::tiles::TileCount(m_w, m_h, kTs);
        uint32_t k = 0;
        for (uint32_t t = 0; t < total; t++) {
            if (t % m_range.stride < m_range.first || t % m_range.stride >= m_range.first + m_range.run) continue;
            for (uint32_t ly = 0; ly < kTs; ly++)
                for (uint32_t lx = 0; lx < kTs; lx++) {
                    const uint32_t x = (t % tx) * kTs + lx, y = (t / tx) * kTs + ly;
                    float* px = o + (size_t(k) * kTs * kTs + ly * kTs + lx) * 4;
                    const bool in = x < m_w && y < m_h;  // padding pixels of edge tiles are zero
                    for (int c = 0; c < 3; c++) px[c] = in ? Pattern(x, y, frame, c) : 0.0f;
                    px[3] = in ? 1.0f : 0.0f;
                }
            k++;
        }
    }
    void RenderFull(uint32_t frame, void* out)
    {
        float* o = static_cast<float*>(out);
        for (uint32_t y = 0; y < m_h; y++)
            for (uint32_t x = 0; x < m_w; x++) {
                for (int c = 0; c < 3; c++) o[(size_t(y) * m_w + x) * 4 + c] = Pattern(x, y, frame, c);
                o[(size_t(y) * m_w + x) * 4 + 3] = 1.0f;
            }
        fullFrames++;
    }
    uint32_t fullFrames = 0;
    void PackRgb(const void* src, uint64_t n, void* dst)
    {
        const float* s = static_cast<const float*>(src);
        float* d = static_cast<float*>(dst);
        for (uint64_t i = 0; i < n; i++) { d[3 * i] = s[4 * i]; d[3 * i + 1] = s[4 * i + 1]; d[3 * i + 2] = s[4 * i + 2]; }
    }
    void UnpackTiles(const void* packed, uint64_t partStridePx, uint32_t nParts, uint32_t first0, uint32_t run, uint32_t stride, void* frame, bool rgb)
    {
        const float* p = static_cast<const float*>(packed);
        float* f = static_cast<float*>(frame);
        const uint32_t tx = (m_w + kTs - 1) / kTs, total = dxrs::tiles::TileCount(m_w, m_h, kTs), ch = rgb ? 3 : 4;
        for (uint32_t part = 0; part < nParts; part++) {
            uint32_t k = 0;
            for (uint32_t t = 0; t < total; t++) {
                const uint32_t res = t % stride;
                if (res < first0 + part * run || res >= first0 + (part + 1) * run) continue;
                for (uint32_t ly = 0; ly < kTs; ly++)
                    for (uint32_t lx = 0; lx < kTs; lx++) {
                        const uint32_t x = (t % tx) * kTs + lx, y = (t / tx) * kTs + ly;
                        if (x >= m_w || y >= m_h) continue;
                        const float* s = p + (size_t(part) * partStridePx + size_t(k) * kTs * kTs + ly * kTs + lx) * ch;
                        float* d = f + (size_t(y) * m_w + x) * 4;
                        d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = rgb ? 1.0f : s[3];
                    }
                k++;
            }
        }
    }
    void Gather(const void* send, void* recv, uint64_t bytes)
    {
        if (m_rank != 0) { m_net.posted[m_rank] = send; return; }
        for (uint32_t r = 1; r < m_world; r++) {
            if (!m_net.posted[r]) { std::fprintf(stderr, "rank %u never posted\n", r); std::exit(3); }
            std::memcpy(static_cast<char*>(recv) + size_t(r - 1) * bytes, m_net.posted[r], bytes);
            m_net.posted[r] = nullptr;
        }
    }

private:
    Network& m_net;
    uint32_t m_rank, m_world, m_w, m_h;
    dxrs::tiles::Range m_range{ 0, 1, 1 };
};

int Check(const float* frame, uint32_t w, uint32_t h, uint32_t frameIndex, const char* what)
{
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            for (int c = 0; c < 4; c++) {
                const float want = c < 3 ? Pattern(x, y, frameIndex, c) : 1.0f, got = frame[(size_t(y) * w + x) * 4 + c];
                if (got != want) { std::fprintf(stderr, "%s: frame %u pixel (%u, %u) channel %d: %g != %g\n", what, frameIndex, x, y, c, got, want); return 1; }
            }
    return 0;
}

int RunCase(uint32_t w, uint32_t h, uint32_t world, uint32_t weight, uint32_t batch, bool rgb)
{
    char what[128];
    std::snprintf(what, sizeof what, "%ux%u world %u weight %u batch %u %s", w, h, world, weight, batch, rgb ? "rgb" : "rgba");
    Network net;
    net.posted.assign(world, nullptr);
    std::vector<std::unique_ptr<FakeBackend>> backends;
    std::vector<std::unique_ptr<dxrs::TileExchange<FakeBackend>>> ranks;
    for (uint32_t r = 0; r < world; r++) {
        backends.push_back(std::make_unique<FakeBackend>(net, r, world, w, h));
        ranks.push_back(std::make_unique<dxrs::TileExchange<FakeBackend>>(*backends[r], w, h, r, world, batch, rgb, kTs));
        ranks[r]->Configure(weight);
    }
    // the partition covers every tile exactly once
    uint32_t tiles = ranks[0]->RootTiles() + (world - 1) * 0;
    for (uint32_t r = 1; r < world; r++) tiles += dxrs::tiles::RangeTileCount(w, h, dxrs::tiles::WeightedPartition(r, world, weight), kTs);
    if (tiles != dxrs::tiles::TileCount(w, h, kTs)) { std::fprintf(stderr, "%s: partition covers %u of %u tiles\n", what, tiles, dxrs::tiles::TileCount(w, h, kTs)); return 1; }
    const uint32_t nFrames = 2 * batch + (batch > 1 ? 1 : 0), first = 40;
    for (uint32_t k = 0; k < nFrames; k++) {
        for (uint32_t r = world; r-- > 0;) ranks[r]->Submit(first + k);  // non-root ranks first: their send is posted when the root gathers
        if ((k + 1) % batch == 0)
            for (uint32_t f = 0; f < batch; f++)
                if (Check(static_cast<const float*>(ranks[0]->Frame(f)), w, h, first + k + 1 - batch + f, what)) return 1;
    }
    for (uint32_t r = world; r-- > 0;) ranks[r]->Finish();
    for (uint32_t f = 0; f < nFrames % batch; f++)
        if (Check(static_cast<const float*>(ranks[0]->Frame(f)), w, h, first + nFrames - nFrames % batch + f, what)) return 1;
    // "do not shard" with several ranks renders whole frames on rank 0 and nothing anywhere else
    const uint32_t wantFull = (world > 1 && weight == 0) ? nFrames : 0u;
    for (uint32_t r = 0; r < world; r++)
        if (backends[r]->fullFrames != (r == 0 ? wantFull : 0u)) { std::fprintf(stderr, "%s: rank %u rendered %u whole frames\n", what, r, backends[r]->fullFrames); return 1; }
    return 0;
}

}  // namespace

int main()
{
    int failed = 0, cases = 0;
    const uint32_t sizes[][2] = { { 100, 70 }, { 64, 64 }, { 33, 65 }, { 257, 31 } };
    for (const auto& sz : sizes)
        for (uint32_t world : { 1u, 2u, 3u, 5u, 8u })
            for (uint32_t weight : { 1u, 2u, 5u, 0u })
                for (uint32_t batch : { 1u, 3u })
                    for (bool rgb : { false, true }) { failed += RunCase(sz[0], sz[1], world, weight, batch, rgb); cases++; }
    std::printf("%d cases, %d failed\n", cases, failed);
    return failed ? 1 : 0;
}
