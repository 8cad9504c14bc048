// raster.hpp -- scanline staging shared by the per-game rasterisers (gfx950, wave64).
//
// One wavefront rasterises one env's frame.  Pixels are produced in row-major 4-pixel groups
// (lane l makes pixels 4l..4l+3 of a scanline), which for RGB is 12 bytes per lane -- a poor
// global-store shape.  RowStager collects UNIT_ROWS scanlines in the wave's private slice of LDS
// and flushes the unit as 16-byte-per-lane stores: every store instruction writes 1 KiB of
// contiguous HBM (measured on MI355X, scripts/ubench/write_bw*.hip: 1 KiB/instr units reach
// ~96 % of the hipMemset rate, 12-byte row stores ~70-85 %).
#pragma once

#include "tbx_common.hpp"
#include "../../include/toybox_amd_spec.h"

// the frame stores are plain 16-byte stores.  Non-temporal (`global_store_dwordx4 ... nt`) stores were measured twice: in
// render-only loops within noise of plain ones (round 2), and in the [step ; render] loops of round 4 -- where a step kernel finds
// its state evicted by the 7.5 GB the rasteriser has just written (profiles/r04_kernel_gaps.txt) and a store that does not
// allocate might have spared it -- SLOWER for large batches: Breakout 1.44 against 1.21 ms per step at 65 536 envs, Amidar 1.58
// against 1.47, SpaceInvaders 2.45 against 2.33; faster only for SpaceInvaders at 4 096 envs (0.152 against 0.161).  Plain stores stay.
__device__ __forceinline__ void tbx_store16(uint4* p, const uint4& v) { *p = v; }

// The 3x5 HUD digit font (bit 3*row + column), looked up out of immediates.  Not a table in memory: inside a rasteriser every
// vector load queues behind the compute unit's own stores (measured on the Breakout kernel: 2 700 cycles for a 64-byte record
// whatever cache holds it), and the `s_waitcnt vmcnt(0)` the compiler puts in front of the first use also waits, on gfx9, for
// every store the wave still has in flight.
__device__ __forceinline__ uint32_t tbx_digit_glyph(uint32_t digit)
{
    constexpr uint16_t G[16] = TBX_DIGIT_FONT;      // ten glyphs, the rest zero
    constexpr uint64_t K0 = G[0] | ((uint64_t)G[1] << 16) | ((uint64_t)G[2] << 32) | ((uint64_t)G[3] << 48);
    constexpr uint64_t K1 = G[4] | ((uint64_t)G[5] << 16) | ((uint64_t)G[6] << 32) | ((uint64_t)G[7] << 48);
    constexpr uint64_t K2 = G[8] | ((uint64_t)G[9] << 16);
    const uint32_t k = digit >> 2;
    const uint64_t w = k == 0 ? K0 : k == 1 ? K1 : k == 2 ? K2 : 0ull;
    return (uint32_t)(w >> (16u * (digit & 3u))) & 0xFFFFu;
}

// The waves a rasteriser launch STARTS with (one per wave slot at five waves per SIMD) sleep for a pseudo-random 0 .. 20 us before
// they paint.
//
// Why: a launch whose first waves all start together into an idle memory system keeps them in lockstep -- every wave composes
// its unit and flushes it at the same moments -- and the frame stores then reach HBM in bursts whose cost depends on where the
// frames lie: the same Breakout loop of [step ; render] ran at 1.20 ms per step into some hipMalloc'ed buffers and at
// 1.34-1.38 ms into others (a process-to-process and box-to-box lottery: the "two rate states" of rounds 2-3;
// scripts/ubench/rate_addr.hip shows it buffer by buffer, profiles/r03_rate_addr.txt).  Back-to-back render launches do not show
// it -- their first waves start against the tail of the previous launch and are spread by that -- and neither did the
// pipelined mode, for the same reason.  With the stagger the loop runs at 1.22 ms into every buffer (at 8 192 envs 0.165-0.168
// instead of 0.176-0.187, at 16 384 0.318-0.322 instead of 0.325-0.357); back to back it costs about half the longest sleep
// once per launch (1.20 against 1.18 ms).  Launches of fewer than 16 384 blocks (a dozen generations of waves) gain nothing and
// are left alone.  SpaceInvaders' RGB launches and Breakout's mid-size ones (up to 65 535 blocks) call it; Breakout's big RGB
// launches reach the same end by going out in two parts (BrkOps::launch_render: 54.0 against 53.5 M env-steps/s at 65 536 envs
// and no slow buffer in 24).  Gray and RGBA launches show no such lottery and only pay for it (1-3 %, same-box A/B), Amidar's
// rasteriser shows none either and loses 6 % with it.
constexpr unsigned TBX_STAGGER_BLOCKS = 1280;      // 256 CUs x 4 SIMDs x 5 wave slots / 4 waves per block
__device__ __forceinline__ void tbx_stagger_first_waves(int wid)
{
    if (blockIdx.x >= TBX_STAGGER_BLOCKS || gridDim.x < 16384u) return;
    const unsigned units = min(12u, gridDim.x / 5120u);
    const int k = (int)((((uint32_t)wid * 2654435761u) >> 26) * units);       // 0 .. 63 * units sleeps of 64 cycles
    for (int i = 0; i < k; i++) __builtin_amdgcn_s_sleep(1);
}

template <int C>
struct PixBytes;   // bytes of a 4-pixel group

template <>
struct PixBytes<4> {
    static __device__ __forceinline__ void write(uint8_t* lds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        *reinterpret_cast<uint4*>(lds) = make_uint4(c0 | 0xFF000000u, c1 | 0xFF000000u, c2 | 0xFF000000u, c3 | 0xFF000000u);
    }
};
template <>
struct PixBytes<3> {
    static __device__ __forceinline__ void write(uint8_t* lds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        c0 &= 0xFFFFFFu; c1 &= 0xFFFFFFu; c2 &= 0xFFFFFFu; c3 &= 0xFFFFFFu;
        uint32_t* p = reinterpret_cast<uint32_t*>(lds);
        p[0] = c0 | (c1 << 24);
        p[1] = (c1 >> 8) | (c2 << 16);
        p[2] = (c2 >> 16) | (c3 << 8);
    }
};
template <>
struct PixBytes<1> {
    static __device__ __forceinline__ void write(uint8_t* lds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        *reinterpret_cast<uint32_t*>(lds) = gray_of(c0) | (gray_of(c1) << 8) | (gray_of(c2) << 16) | (gray_of(c3) << 24);
    }
};

// a palette colour in the form put4p() takes: the gray byte for C == 1, RGBA otherwise (convert once, not per pixel)
template <int C>
__device__ __forceinline__ uint32_t pix_of(uint32_t rgba) { return C == 1 ? gray_of(rgba) : rgba; }

// rows [a, b) of a frame as bits of 64-row word k (rows outside 0..255 fall away)
__device__ __forceinline__ uint64_t row_range_bits(long a, long b, int k)
{
    const long lo = a > 64L * k ? a : 64L * k, hi = b < 64L * k + 64 ? b : 64L * k + 64;
    if (hi <= lo) return 0ull;
    const int n = (int)(hi - lo), sh = (int)(lo - 64L * k);
    return (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << sh;
}

// m[w] for a run-time w by VALUE (a conditional over array elements is a select of addresses, which pushes the
// enclosing object into scratch)
__device__ __forceinline__ uint64_t sel4(int w, uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return w == 0 ? a : w == 1 ? b : w == 2 ? c : d; }

// R consecutive row bits starting at row `pos` of a 256-row mask held as four wave-uniform words
template <int R>
__device__ __forceinline__ uint32_t row_mask_chunk(const uint64_t (&m)[4], int pos)
{
    const int w = pos >> 6, b = pos & 63;
    const uint64_t cur = sel4(w, m[0], m[1], m[2], m[3]);
    const uint64_t nxt = sel4(w, m[1], m[2], m[3], 0ull);
    uint64_t v = cur >> b;
    if (b > 64 - R) v |= nxt << (64 - b);
    return (uint32_t)v & ((1u << R) - 1u);
}

// W: frame width (multiple of 4), C: channels, R: scanlines per unit.
template <int C, int W, int R>
struct RowStager {
    static constexpr int ROW_BYTES = W * C;
    static constexpr int UNIT_BYTES = ROW_BYTES * R;
    static constexpr int GROUPS = W / 4;                    // 4-pixel groups per scanline
    static constexpr int GROUP_ITERS = (GROUPS + 63) / 64;  // passes a wave needs per scanline
    static_assert(ROW_BYTES % 16 == 0, "scanlines must be a whole number of 16-byte chunks");

    uint8_t* lds;   // this wave's UNIT_BYTES slice, 16-byte aligned

    __device__ __forceinline__ void put4(int row_in_unit, int group, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const
    {
        PixBytes<C>::write(lds + row_in_unit * ROW_BYTES + group * 4 * C, c0, c1, c2, c3);
    }

    // the same for colours that went through pix_of<C>() already
    __device__ __forceinline__ void put4p(int row_in_unit, int group, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const
    {
        if (C == 1)
            *reinterpret_cast<uint32_t*>(lds + row_in_unit * ROW_BYTES + group * 4) = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        else
            PixBytes<C>::write(lds + row_in_unit * ROW_BYTES + group * 4 * C, c0, c1, c2, c3);
    }

    // a whole unit of one colour (already through pix_of<C>()) straight to dst, without staging.  RGB: the byte stream
    // has period 3, a 16-byte chunk c starts at phase c % 3 and holds the dwords d[ph], d[ph+1], d[ph+2], d[ph]
    static __device__ __forceinline__ void fill_unit(uint8_t* dst, int lane, uint32_t p)
    {
        uint4* out = reinterpret_cast<uint4*>(dst);
        constexpr int chunks = UNIT_BYTES / 16;
        if (C == 3) {
            const uint32_t q = p & 0xFFFFFFu;
            const uint32_t d0 = q | (q << 24), d1 = (q >> 8) | (q << 16), d2 = (q >> 16) | (q << 8);
            const int ph0 = lane % 3;
#pragma unroll
            for (int i = 0; i < (chunks + 63) / 64; i++) {
                const int c = lane + 64 * i;
                const int ph = (ph0 + i) % 3;               // (lane + 64 i) % 3
                const uint4 v = ph == 0 ? make_uint4(d0, d1, d2, d0) : ph == 1 ? make_uint4(d1, d2, d0, d1) : make_uint4(d2, d0, d1, d2);
                if (c < chunks) tbx_store16(out + c, v);
            }
        } else {
            const uint32_t q = C == 1 ? p * 0x01010101u : (p | 0xFF000000u);
            const uint4 v = make_uint4(q, q, q, q);
#pragma unroll
            for (int i = 0; i < (chunks + 63) / 64; i++) {
                const int c = lane + 64 * i;
                if (c < chunks) tbx_store16(out + c, v);
            }
        }
    }

    // Store ADDRESSES (round 5).  Units whose size is no multiple of 128 bytes (Amidar: 10 scanlines x 480 B
    // = 4 800 B = 64 x 75) start, every other one, 64 bytes into a 128-byte line, and a 1-KiB store instruction that starts
    // there touches nine lines instead of eight.  As a PURE store stream that costs 8-11 % (scripts/ubench/write_align.hip,
    // profiles/r05_write_align.txt: every unit moved by 64 bytes; 15 % for 16 bytes; Amidar's geometry +9 % with 128-byte
    // aligned unit starts), while the alignment of the FRAMES -- dense, or strides rounded up to 128 B .. 4 KiB -- changes
    // nothing (+-1 %).  Inside the rasteriser it does not carry over: a flush whose lane -> chunk mapping is counted from the
    // 128-byte line dst lies in (every instruction covers whole lines, the chunks in front of dst masked off, 38 instead of 42
    // line requests per misaligned unit, same instruction count) was parity-green and 2.7 % SLOWER for Amidar RGB (1.412
    // against 1.371 ms at 65 536 envs) and gray (0.685 / 0.667), within noise at 4 096 (profiles/r05_experiments.txt) --
    // removed.  The rasterisers' store rate is address-independent; what bounds it is in DESIGN.md section 6.

    // write `rows` staged scanlines to dst (16-byte aligned, contiguous)
    __device__ __forceinline__ void flush(uint8_t* dst, int lane, int rows = R) const
    {
        const int chunks = rows * (ROW_BYTES / 16);
        const uint4* src = reinterpret_cast<const uint4*>(lds);
        uint4* out = reinterpret_cast<uint4*>(dst);
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int i = lane; i < chunks; i += 64) tbx_store16(out + i, src[i]);
        __builtin_amdgcn_wave_barrier();
    }
};
