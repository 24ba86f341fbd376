/*
 * TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see oracle/README.md).
 *
 * ImageTransform.__call__ (mmdet/datasets/transforms.py:31-50) =
 *   mmcv.imrescale / imresize -> mmcv.imnormalize -> mmcv.imflip -> mmcv.impad_to_multiple ->
 *   HWC -> CHW.
 * PARITY UNPINNED for the resize step: the pixels come from two third-party packages that are
 * absent from /root/reference and from this image -- mmcv (INSTALL.md pins mmcv 0.2.x; its
 * imrescale / imresize / imnormalize / imflip / impad_to_multiple are restated from their
 * published source) and OpenCV (cv2.resize, INTER_LINEAR, 8-bit: restated from the published
 * resize.cpp -- 11-bit fixed-point coefficients, INTER_RESIZE_COEF_BITS = 11).  There is no
 * golden vector for it in the reference (it ships no tests) and cv2 cannot be run here, so
 * the resize is checked only against an exact-arithmetic bilinear (within 1 grey level);
 * everything after the resize (normalise, flip, pad, transpose) is exact IEEE fp32 and is
 * pinned against numpy.
 *
 * cv2.resize(src 8UC3, dsize, INTER_LINEAR), as restated:
 *   scale_x = 1.0 / ((double)dw / sw);  fx = (float)((dx + 0.5) * scale_x - 0.5);
 *   sx = floor(fx); fx -= sx;  sx < 0 -> (sx, fx) = (0, 0);  sx >= sw-1 -> (sw-1, 0)
 *   alpha = { round_half_even((1.f - fx) * 2048), round_half_even(fx * 2048) }   (short)
 *   same for y, except that the ROW indices sy, sy+1 are clamped to [0, sh-1], not fy
 *   H[k][dx] = S[sy_k][sx] * a0 + S[sy_k][sx+1] * a1          (sx+1 >= sw: S[sx] * 2048)
 *   dst = (uint8)(( ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2 ) >> 2)
 *   dsize == ssize: copy.  (The exact 2x reduction takes cv2's INTER_AREA fast path, which
 *   gives the same bytes as the formula above.)                                           */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static short pp_coef(float v)                 /* saturate_cast<short>(v): cvRound, half to even */
{
    double r = nearbyint((double)v);
    if (r > 32767.0) r = 32767.0;
    if (r < -32768.0) r = -32768.0;
    return (short)r;
}

static void pp_axis(int dn, int sn, int clamp_frac, int *ofs, short *coef)
{
    const double scale = 1.0 / ((double)dn / (double)sn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (clamp_frac) {
            if (s < 0) { s = 0; f = 0.0f; }
            if (s >= sn - 1) { s = sn - 1; f = 0.0f; }
        }
        ofs[d] = s;
        coef[2 * d] = pp_coef((1.0f - f) * 2048.0f);
        coef[2 * d + 1] = pp_coef(f * 2048.0f);
    }
}

/* src (sh, sw, 3) u8 -> dst (dh, dw, 3) u8 */
void ia_o_resize_bilinear_u8(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh, int dw)
{
    if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * 3); return; }
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    short *xa = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    short *yb = (short *)malloc(sizeof(short) * 2 * (size_t)dh);
    pp_axis(dw, sw, 1, xofs, xa);
    pp_axis(dh, sh, 0, yofs, yb);
    for (int dy = 0; dy < dh; ++dy) {
        int r0 = yofs[dy], r1 = yofs[dy] + 1;
        r0 = r0 < 0 ? 0 : (r0 >= sh ? sh - 1 : r0);
        r1 = r1 < 0 ? 0 : (r1 >= sh ? sh - 1 : r1);
        const int b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx], a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
            const int two = sx + 1 < sw;
            for (int c = 0; c < 3; ++c) {
                const uint8_t *p0 = src + ((size_t)r0 * sw + sx) * 3 + c;
                const uint8_t *p1 = src + ((size_t)r1 * sw + sx) * 3 + c;
                const int h0 = two ? p0[0] * a0 + p0[3] * a1 : p0[0] * 2048;
                const int h1 = two ? p1[0] * a0 + p1[3] * a1 : p1[0] * 2048;
                dst[((size_t)dy * dw + dx) * 3 + c] =
                    (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs); free(yofs); free(xa); free(yb);
}

/* imnormalize (to_rgb: BGR -> RGB first) + imflip (horizontal) + impad_to_multiple (zeros at
 * the bottom / right) + transpose(2,0,1): img (h, w, 3) u8 -> out (3, ph, pw) fp32.       */
void ia_o_normalize_flip_pad_chw(const uint8_t *img, int h, int w, const float *mean,
                                 const float *std, int to_rgb, int flip, int ph, int pw,
                                 float *out)
{
    memset(out, 0, sizeof(float) * 3 * (size_t)ph * pw);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sx = flip ? (w - 1 - x) : x;
            for (int c = 0; c < 3; ++c) {
                const int sc = to_rgb ? 2 - c : c;
                const float v = (float)img[((size_t)y * w + sx) * 3 + sc];
                out[((size_t)c * ph + y) * pw + x] = (v - mean[c]) / std[c];
            }
        }
}
