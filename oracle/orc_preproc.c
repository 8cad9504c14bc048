/* orc_preproc.c -- CPU restatement of the agent-side wrapper stack.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows the reference's vendored baselines: MaxAndSkipEnv (baselines/baselines/common/atari_wrappers.py:193-219),
 * WarpFrame (:230-244; Toybox frames are already gray, :241-242), ClipRewardEnv (:221-227) and VecFrameStack
 * (common/vec_env/vec_frame_stack.py:17-30), with the VecEnv auto-reset (dummy_vec_env.py:51-54).
 * PARITY UNPINNED against the reference here: WarpFrame calls cv2.resize(INTER_AREA), and OpenCV is neither in the
 * reference tree nor installed; this file states INTER_AREA's definition (area-weighted mean) in exact integer
 * arithmetic with round-half-up.  Two documented simplifications: when the game ends inside the `skip` frames the
 * observation is the warped reset frame (what VecEnv returns), so MaxAndSkipEnv's stale buffer never shows. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* dst[oy][ox] = round( area-weighted mean of src over [ox*W/ow,(ox+1)*W/ow) x [oy*H/oh,(oy+1)*H/oh) ) */
void orc_warp_area(const uint8_t* src, int H, int W, uint8_t* dst, int oh, int ow)
{
    for (int oy = 0; oy < oh; oy++)
        for (int ox = 0; ox < ow; ox++) {
            /* work in a grid refined by (oh, ow): source pixel = oh x ow cells, output pixel = H x W cells */
            const long y0 = (long)oy * H, y1 = (long)(oy + 1) * H, x0 = (long)ox * W, x1 = (long)(ox + 1) * W;
            long sum = 0;
            for (long sy = y0 / oh; sy * oh < y1; sy++) {
                const long ya = sy * oh > y0 ? sy * oh : y0, yb = (sy + 1) * oh < y1 ? (sy + 1) * oh : y1;
                for (long sx = x0 / ow; sx * ow < x1; sx++) {
                    const long xa = sx * ow > x0 ? sx * ow : x0, xb = (sx + 1) * ow < x1 ? (sx + 1) * ow : x1;
                    sum += (yb - ya) * (xb - xa) * (long)src[sy * W + sx];
                }
            }
            const long area = (long)H * W;
            dst[oy * ow + ox] = (uint8_t)((sum + area / 2) / area);
        }
}

/* roll the stack of one env and write the new frame last; fresh = 1: zero the older slots first (VecFrameStack,
 * vec_frame_stack.py:22-33), fresh = 2: the older slots get the new frame too (FrameStack.reset, atari_wrappers.py:257-261) */
void orc_stack_push(uint8_t* obs, const uint8_t* frame, int oh, int ow, int stack, int fresh)
{
    for (int p = 0; p < oh * ow; p++) {
        uint8_t* px = obs + (size_t)p * stack;
        for (int c = 0; c + 1 < stack; c++) px[c] = fresh == 2 ? frame[p] : fresh ? 0 : px[c + 1];
        px[stack - 1] = frame[p];
    }
}
