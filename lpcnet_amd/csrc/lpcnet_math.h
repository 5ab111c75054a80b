// Scalar building blocks of the sample loop, shared by the HIP kernels (device) and by the
// host-side C++ unit hooks.  Every function reproduces, operation for operation, the
// arithmetic of the reference's *generic-C* flavour, which is the bit-reproducible one
// (SURVEY.md fact 8).  The translation unit MUST be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define LPCN_HD __host__ __device__ __forceinline__
#else
#define LPCN_HD static inline
#endif

// tanh: nearest knot of a 201-entry table (spacing 0.04) + one correction step.
// reference: src/vec.h:82-99.  `tab` may live in LDS, constant or host memory.
LPCN_HD float lpcn_tanh(float x, const float *tab)
{
    const float ax = fabsf(x);
    int i = (int)(.5f + 25.f * ax);                // == (int)floor(.5 + 25*ax) of the reference: the argument is >= 0.5, where truncation IS floor (one VALU instruction less)
    i = i > 200 ? 200 : i;                         // i >= 0 always since ax >= 0
    const float dx = ax - .04f * (float)i;
    const float y = tab[i];
    const float dy = 1.f - y * y;
    const float r = y + dx * dy * (1.f - y * dx);
    return x < 0 ? -r : r;                         // sign*y with sign = +-1 is an exact sign flip
}

// reference: src/vec.h:101-104
LPCN_HD float lpcn_sigmoid(float x, const float *tab)
{
    return .5f + .5f * lpcn_tanh(.5f * x, tab);
}

// reference: src/common.h:18-33
LPCN_HD float lpcn_log2(float x)
{
    union { float f; int32_t i; } in;
    in.f = x;
    const int integer = (in.i >> 23) - 127;
    in.i -= integer << 23;
    float frac = in.f - 1.5f;
    frac = -0.41445418f + frac * (0.95909232f + frac * (-0.33951290f + frac * 0.16541097f));
    return (float)(1 + integer) + frac;
}

// reference: src/common.h:47-58.  All float except the final floor(.5 + u) which the
// reference evaluates in double; u is in [0,255] so the double add is exact.
LPCN_HD int lpcn_lin2ulaw(float x)
{
    const float scale = 255.f / 32768.f;
    const float s = x >= 0 ? 1.f : -1.f;
    x = fabsf(x);
    float u = (s * (128.f * (0.69315f * lpcn_log2(1.f + scale * x)) / 5.5451774445f));
    u = 128.f + u;
    if (u < 0) u = 0;
    if (u > 255) u = 255;
    return (int)floor(.5 + (double)u);
}

// reference: src/kiss99.c:59-81; state = {z, w, jsr, jcong}
LPCN_HD uint32_t lpcn_kiss99(uint32_t *c)
{
    const uint32_t znew = 36969u * (c[0] & 0xFFFFu) + (c[0] >> 16);
    const uint32_t wnew = 18000u * (c[1] & 0xFFFFu) + (c[1] >> 16);
    const uint32_t mwc = (znew << 16) + wnew;
    uint32_t jsr = c[2];
    jsr ^= jsr << 13;
    jsr ^= jsr >> 17;
    jsr ^= jsr << 5;
    const uint32_t cong = 69069u * c[3] + 1234567u;
    c[0] = znew; c[1] = wnew; c[2] = jsr; c[3] = cong;
    return (mwc ^ cong) + jsr;
}

// reference: src/lpcnet.c:265-269 -- de-emphasised sample to int16 (double floor like the C)
LPCN_HD int lpcn_round_pcm(float pcm)
{
    if (pcm < -32767.f) pcm = -32767.f;
    if (pcm > 32767.f) pcm = 32767.f;
    return (int)floor(.5 + (double)pcm);
}
