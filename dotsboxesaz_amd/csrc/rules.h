// rules.h -- Dots & Boxes rules on an edge bitmask (device + host inline).
// Restates BoxesState (reference dots_boxes/dots_boxes_game.py:30-109) on the SoA
// representation: edges[4] = bitmask of played edges (bit index = action index),
// b2c = 2*boxes_to_close (integers), to_play, just_played (-1 = None).
#pragma once

#include "common.h"

struct GState {
    uint64_t e0, e1, e2, e3;
    int b2c0, b2c1;
    int to_play, just_played;
};

#define HD __host__ __device__ __forceinline__

HD uint64_t gs_word(const GState &s, int w)
{
    return w == 0 ? s.e0 : (w == 1 ? s.e1 : (w == 2 ? s.e2 : s.e3));
}

HD bool gs_bit(const GState &s, int idx)
{
    return (gs_word(s, idx >> 6) >> (idx & 63)) & 1ull;
}

HD void gs_set(GState &s, int idx)
{
    uint64_t b = 1ull << (idx & 63);
    int w = idx >> 6;
    s.e0 |= (w == 0) ? b : 0ull;
    s.e1 |= (w == 1) ? b : 0ull;
    s.e2 |= (w == 2) ? b : 0ull;
    s.e3 |= (w == 3) ? b : 0ull;
}

HD bool geo_sentinel(const Geo &g, int idx)
{
    int w = idx >> 6;
    uint64_t m = w == 0 ? g.sentinel[0] : (w == 1 ? g.sentinel[1] : (w == 2 ? g.sentinel[2] : g.sentinel[3]));
    return (m >> (idx & 63)) & 1ull;
}

// BoxesState.__init__, dots_boxes_game.py:30-39
HD void gs_init(const Geo &g, GState &s)
{
    s.e0 = s.e1 = s.e2 = s.e3 = 0ull;
    s.b2c0 = g.B;
    s.b2c1 = g.B;
    s.to_play = 0;
    s.just_played = -1;
}

// get_valid_moves, dots_boxes_game.py:44-49: board.ravel()[i] == 0
HD bool gs_valid(const Geo &g, const GState &s, int idx)
{
    return idx >= 0 && idx < g.A && !gs_bit(s, idx) && !geo_sentinel(g, idx);
}

// get_result, dots_boxes_game.py:51-59 (perspective of to_play)
HD int gs_result(const GState &s)
{
    int own = s.to_play == 0 ? s.b2c0 : s.b2c1;
    int opp = s.to_play == 0 ? s.b2c1 : s.b2c0;
    if (s.b2c0 == 0 && s.b2c1 == 0)
        return 0;
    if (own < 0)
        return 1;
    if (opp < 0)
        return -1;
    return DBAZ_RESULT_NONE;
}

// _check_box, dots_boxes_game.py:102-104
HD bool gs_box(const Geo &g, const GState &s, int l, int c)
{
    int h0 = l * g.W + c, h1 = (l + 1) * g.W + c;
    int v0 = g.HW + l * g.W + c, v1 = v0 + 1;
    return gs_bit(s, h0) && gs_bit(s, h1) && gs_bit(s, v0) && gs_bit(s, v1);
}

// play_, dots_boxes_game.py:61-89.  Returns #closed boxes or -1 (illegal, state untouched).
// closed (may be null) receives up to two (l, c) pairs.
HD int gs_play(const Geo &g, GState &s, int move, int *closed)
{
    if (!gs_valid(g, s, move))
        return -1;
    gs_set(s, move);
    int p = move / g.HW;
    int rem = move - p * g.HW;
    int l = rem / g.W;
    int c = rem - l * g.W;
    int n = 0;
    if (p == 0) {
        if (l > 0 && gs_box(g, s, l - 1, c)) {
            if (closed) { closed[2 * n] = l - 1; closed[2 * n + 1] = c; }
            n++;
        }
        if (l < g.H - 1 && gs_box(g, s, l, c)) {
            if (closed) { closed[2 * n] = l; closed[2 * n + 1] = c; }
            n++;
        }
    } else {
        if (c > 0 && gs_box(g, s, l, c - 1)) {
            if (closed) { closed[2 * n] = l; closed[2 * n + 1] = c - 1; }
            n++;
        }
        if (c < g.W - 1 && gs_box(g, s, l, c)) {
            if (closed) { closed[2 * n] = l; closed[2 * n + 1] = c; }
            n++;
        }
    }
    s.just_played = s.to_play;
    if (n == 0) {
        s.to_play = 1 - s.to_play;
    } else {
        if (s.to_play == 0) s.b2c0 -= 2 * n; else s.b2c1 -= 2 * n;
    }
    return n;
}

// get_features, dots_boxes_game.py:96-100: element i of the [3,H,W] ravel
HD int gs_feature(const Geo &g, const GState &s, int i)
{
    if (i < 2 * g.HW)
        return gs_bit(s, i) ? 1 : 0; // board // 255 (sentinels are 1 // 255 = 0)
    int v = s.to_play == 0 ? s.b2c0 : s.b2c1;
    return (int)(int8_t)v; // np.int8(2 * boxes_to_close[to_play])
}

HD int gs_count_valid(const Geo &g, const GState &s)
{
    int n = 0;
    uint64_t w[4] = {s.e0, s.e1, s.e2, s.e3};
    for (int k = 0; k < 4; k++) {
        uint64_t free_ = ~(w[k] | g.sentinel[k]) & g.amask[k];
#if defined(__HIP_DEVICE_COMPILE__)
        n += __popcll(free_);
#else
        n += __builtin_popcountll(free_);
#endif
    }
    return n;
}
