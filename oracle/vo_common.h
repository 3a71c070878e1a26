/*
 * vo_common.h -- shared helpers for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * The reference (nomadoor/ComfyUI-Video-Stabilizer) is pure Python; all of its
 * pixel arithmetic is delegated to OpenCV (opencv-python-headless>=4.8,<5,
 * pyproject.toml:8 -- an unpinned range, not vendored, absent from this
 * container).  The functions in oracle/ restate the published OpenCV 4.x
 * algorithms behind the reference's call sites.  PARITY UNPINNED against real
 * OpenCV: no cv2 exists here and the reference ships no numeric golden vectors
 * (SURVEY.md section 8c); what pins the oracle is listed in DESIGN.md.
 *
 * Build: -ffp-contract=off is mandatory (OpenCV's baseline x86-64 build has no
 * FMA in these loops, and the HIP kernels are built the same way so results can
 * be compared bit for bit).
 */
#ifndef VO_COMMON_H
#define VO_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cvRound(double): round half to even (lrint under the default rounding mode). */
static inline int vo_round_d(double v) { return (int)lrint(v); }
static inline int vo_round_f(float v) { return (int)lrintf(v); }
static inline int vo_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int vo_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int vo_ceil_d(double v) { int i = (int)v; return i + (i < v); }

static inline short vo_sat_short(int v)
{
    return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}
static inline unsigned char vo_sat_u8_i(int v)
{
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
/* saturate_cast<uchar>(float) = saturate(cvRound(v)) */
static inline unsigned char vo_sat_u8_f(float v) { return vo_sat_u8_i(vo_round_f(v)); }

#ifdef __cplusplus
}
#endif
#endif
