// TileExchange.hpp -- the multi-GPU frame exchange for a C++ host (BASELINE north_star: "Host stays C++ ... RCCL gather
// of HDR tiles over xGMI"): one process per GPU, the frame cut into 32x32 tiles, every rank renders its residue range of the
// tiles (pt_set_partition_ex / pt_render_tiles), the packed tiles are gathered to rank 0 (pt_gather) and un-swizzled there
// (pt_unpack_tiles_ex / _rgb).  The reference renders on one adapter (Source/DeviceResources.cpp:507); this is the path's
// multi-GPU extension, the C++ twin of directx-raytracing-spheres-demo_amd/exchange.py (same partition, same buffers, same order
// of calls -- the Python file documents the reasoning: root-weighted partition, one collective per batch of frames, 12-byte
// pixels on the links).
//
// The control flow is written against a Backend so that it can be exercised without GPUs (tests/cpp/tile_exchange_test.cpp drives
// it with host memory and an in-process gather); PtBackend is the real one: the C-ABI of include/pt_api.h.
#pragma once

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <utility>
#include <vector>

#include "Raytracing.hpp"

namespace dxrs {

namespace tiles {

struct Range { uint32_t first, run, stride; };  // the tiles t with first <= t % stride < first + run, in increasing t

inline uint32_t TileCount(uint32_t w, uint32_t h, uint32_t ts = 32) { return ((w + ts - 1) / ts) * ((h + ts - 1) / ts); }

// rank 0 (which assembles the frame, so its tiles need no transfer) carries rootWeight shares, every other rank one;
// rootWeight == 0: rank 0 renders everything (with several ranks: whole frames through Backend::RenderFull, not tiles)
inline Range WeightedPartition(uint32_t rank, uint32_t world, uint32_t rootWeight)
{
    if (world == 1 || rootWeight == 0) return rank == 0 ? Range{ 0, 1, 1 } : Range{ 0, 0, 1 };
    const uint32_t stride = world - 1 + rootWeight;
    return rank == 0 ? Range{ 0, rootWeight, stride } : Range{ rootWeight - 1 + rank, 1, stride };
}

inline uint32_t RangeTileCount(uint32_t w, uint32_t h, Range r, uint32_t ts = 32)
{
    const uint32_t total = TileCount(w, h, ts), rem = total % r.stride;
    return (total / r.stride) * r.run + (rem > r.first ? (rem - r.first < r.run ? rem - r.first : r.run) : 0u);
}

}  // namespace tiles

// The real backend: device memory and operations through the C-ABI.  setFrame(k) installs frame k's camera and constants
// (pt_set_camera / pt_set_constants) before the rank renders its tiles of that frame.
class PtBackend {
public:
    PtBackend(PtContext* ctx, std::function<void(uint32_t)> setFrame) : m_ctx(ctx), m_setFrame(std::move(setFrame)) {}

    void* Alloc(size_t bytes)
    {
        void* p = nullptr;
        ThrowIfFailed(pt_device_alloc(m_ctx, bytes ? bytes : 16, &p), m_ctx, "pt_device_alloc");
        return p;
    }
    void Free(void* p) noexcept { (void)pt_device_free(m_ctx, p); }
    void SetPartition(tiles::Range r) { ThrowIfFailed(pt_set_partition_ex(m_ctx, r.first, r.run, r.stride), m_ctx, "pt_set_partition_ex"); }
    void RenderTiles(uint32_t frameIndex, void* out)
    {
        m_setFrame(frameIndex);
        ThrowIfFailed(pt_render_tiles(m_ctx, out, nullptr), m_ctx, "pt_render_tiles");
    }
    void RenderFull(uint32_t frameIndex, void* frame)  // the whole frame, row-major float4: root weight 0 ("do not shard")
    {
        m_setFrame(frameIndex);
        ThrowIfFailed(pt_render(m_ctx, nullptr, frame, 1, nullptr), m_ctx, "pt_render");
    }
    void PackRgb(const void* src, uint64_t nPixels, void* dst) { ThrowIfFailed(pt_pack_rgb(m_ctx, src, nPixels, dst), m_ctx, "pt_pack_rgb"); }
    void UnpackTiles(const void* packed, uint64_t partStridePx, uint32_t nParts, uint32_t first0, uint32_t run, uint32_t stride, void* frame, bool rgb)
    {
        ThrowIfFailed((rgb ? pt_unpack_tiles_rgb : pt_unpack_tiles_ex)(m_ctx, packed, partStridePx, nParts, first0, run, stride, frame), m_ctx, "pt_unpack_tiles");
    }
    void Gather(const void* send, void* recv, uint64_t bytes) { ThrowIfFailed(pt_gather(m_ctx, send, recv, bytes, 0), m_ctx, "pt_gather"); }

private:
    PtContext* m_ctx;
    std::function<void(uint32_t)> m_setFrame;
};

// Frame exchange of one rank.  Every rank constructs it with the same (w, h, world, batch, rgb) and calls Configure with the same
// root weight; then Submit(k) per frame in the same order on every rank, Finish() at the end.  Rank 0 finds frame k of the
// current batch in Frame(k % batch) once the batch has been flushed (device memory, ordered on the backend's stream).
template <class Backend>
class TileExchange {
public:
    // framesInFlight: PtConfig::frames_in_flight of the context that renders (0 / 1 = one).  The two batch buffers alternate, and a frame
    // only waits for the consumer of ITS buffer from framesInFlight - 1 calls back (pt_api.hip render_common): a batch shorter than
    // that would let a frame overwrite tiles the gather / un-swizzle of the previous batch has not read yet -- refused here.
    TileExchange(Backend& backend, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t batch, bool rgb = true, uint32_t ts = 32, uint32_t framesInFlight = 1)
        : m_b(backend), m_w(w), m_h(h), m_rank(rank), m_world(world), m_batch(batch ? batch : 1), m_ts(ts), m_rgb(rgb), m_tilePx(uint64_t(ts) * ts)
    {
        if (world == 0 || rank >= world) throw std::invalid_argument("TileExchange: need rank < world");
        if (framesInFlight > 1 && m_batch + 1 < framesInFlight) throw std::invalid_argument("TileExchange: batch must be at least frames in flight - 1");
        // buffers sized for the largest share any root weight can give this rank
        const uint64_t capRoot = tiles::TileCount(w, h, ts);
        m_capOther = world > 1 ? tiles::RangeTileCount(w, h, tiles::Range{ 1, 1, world }, ts) : 0;
        const uint64_t capOwn = rank == 0 ? capRoot : m_capOther;
        for (auto& p : m_own) p = m_b.Alloc(m_batch * capOwn * m_tilePx * 16);
        const uint64_t ch = rgb ? 12 : 16;
        if (rank == 0) {
            for (uint32_t f = 0; f < m_batch; f++) m_frames.push_back(m_b.Alloc(uint64_t(w) * h * 16));
            if (world > 1) m_gathered = m_b.Alloc((world - 1) * m_batch * m_capOther * m_tilePx * ch);
        } else if (rgb && world > 1) {
            m_send = m_b.Alloc(m_batch * m_capOther * m_tilePx * 12);
        }
        Configure(1);
    }
    ~TileExchange()
    {
        for (auto p : m_own) m_b.Free(p);
        for (auto p : m_frames) m_b.Free(p);
        if (m_gathered) m_b.Free(m_gathered);
        if (m_send) m_b.Free(m_send);
    }
    TileExchange(const TileExchange&) = delete;
    TileExchange& operator=(const TileExchange&) = delete;

    // select the partition (between batches; all ranks the same weight).  Nothing is reallocated.
    void Configure(uint32_t rootWeight)
    {
        if (m_submitted % m_batch) throw std::logic_error("TileExchange::Configure between batches only");
        m_rootWeight = rootWeight;
        m_range = tiles::WeightedPartition(m_rank, m_world, rootWeight);
        m_sharded = m_world > 1 && rootWeight != 0;
        m_direct = m_world > 1 && rootWeight == 0;  // rank 0 renders whole frames straight into Frame(slot): no tiles, no un-swizzle
        m_nRoot = tiles::RangeTileCount(m_w, m_h, tiles::WeightedPartition(0, m_world, rootWeight), m_ts);
        // every non-root rank sends the same number of tiles (the first of them owns the most; later ones are zero padded)
        m_nOther = m_sharded ? tiles::RangeTileCount(m_w, m_h, tiles::WeightedPartition(1, m_world, rootWeight), m_ts) : 0;
        m_ownPx = uint64_t(m_rank == 0 ? m_nRoot : m_nOther) * m_tilePx;
        m_otherPx = uint64_t(m_nOther) * m_tilePx;
        m_b.SetPartition(m_range);
    }

    // queue one frame; the collective + un-swizzle are issued when its batch is complete
    void Submit(uint32_t frameIndex)
    {
        const uint32_t b = (m_submitted / m_batch) % 2, slot = m_submitted % m_batch;
        if (m_direct) { if (m_rank == 0) m_b.RenderFull(frameIndex, m_frames[slot]); }
        else if (m_ownPx) m_b.RenderTiles(frameIndex, static_cast<char*>(m_own[b]) + uint64_t(slot) * m_ownPx * 16);
        m_submitted++;
        if (slot == m_batch - 1) Flush(b, m_batch);
    }

    // flush a partially filled batch (end of the run)
    void Finish()
    {
        const uint32_t pending = m_submitted % m_batch;
        if (pending) {
            Flush((m_submitted / m_batch) % 2, pending);
            m_submitted += m_batch - pending;  // the next Submit starts a fresh batch
        }
    }

    void* Frame(uint32_t slot) const { return m_frames.at(slot); }  // rank 0: W * H float4, device memory
    uint32_t RootWeight() const { return m_rootWeight; }
    uint32_t RootTiles() const { return m_nRoot; }
    uint32_t TilesPerOtherRank() const { return m_nOther; }

private:
    void Flush(uint32_t b, uint32_t nFrames)
    {
        if (m_direct) return;
        const uint64_t px = m_rgb ? 12 : 16;
        if (m_sharded) {
            // the whole batch buffer travels (a partial last batch leaves its tail unused): every rank sends the same byte count
            const uint64_t bytes = uint64_t(m_batch) * m_otherPx * px;
            if (m_rank == 0) {
                m_b.Gather(nullptr, m_gathered, bytes);
            } else {
                const void* send = m_own[b];
                if (m_rgb) { m_b.PackRgb(m_own[b], uint64_t(m_batch) * m_ownPx, m_send); send = m_send; }
                m_b.Gather(send, nullptr, bytes);
            }
        }
        if (m_rank != 0) return;
        for (uint32_t f = 0; f < nFrames; f++) {
            // the root's own range straight from where it was rendered, then the gathered ranges of ranks 1 .. world - 1
            m_b.UnpackTiles(static_cast<char*>(m_own[b]) + uint64_t(f) * m_ownPx * 16, 0, 1, m_range.first, m_range.run, m_range.stride, m_frames[f], false);
            if (m_sharded)
                m_b.UnpackTiles(static_cast<char*>(m_gathered) + uint64_t(f) * m_otherPx * px, uint64_t(m_batch) * m_otherPx, m_world - 1, m_range.run, 1,
                                m_range.stride, m_frames[f], m_rgb);
        }
    }

    Backend& m_b;
    uint32_t m_w, m_h, m_rank, m_world, m_batch, m_ts;
    bool m_rgb;
    uint64_t m_tilePx, m_capOther = 0;
    void* m_own[2] = { nullptr, nullptr };  // two batch buffers alternate: the next batch renders while the previous one is on the links
    std::vector<void*> m_frames;
    void* m_gathered = nullptr;
    void* m_send = nullptr;
    uint32_t m_rootWeight = 1, m_nRoot = 0, m_nOther = 0;
    tiles::Range m_range{ 0, 1, 1 };
    bool m_sharded = false, m_direct = false;
    uint64_t m_ownPx = 0, m_otherPx = 0, m_submitted = 0;
};

}  // namespace dxrs
