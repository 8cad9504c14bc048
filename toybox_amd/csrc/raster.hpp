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

    // write `rows` staged scanlines to dst (16-byte aligned, contiguous)
    __device__ __forceinline__ void flush(uint8_t* dst, int lane, int rows = R) const
    {
        const int chunks = rows * (ROW_BYTES / 16);
        const uint4* src = reinterpret_cast<const uint4*>(lds);
        uint4* out = reinterpret_cast<uint4*>(dst);
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int i = lane; i < chunks; i += 64) out[i] = src[i];
        __builtin_amdgcn_wave_barrier();
    }
};
