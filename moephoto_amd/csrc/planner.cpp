// planner.cpp -- host-side tile planner: which overlapping tiles doCrop cuts an image into, and the per-axis
// cover tables the stitch kernel folds over.  Pure index arithmetic, no device.
//
// Reference: getAnchors (python/imageProcess.py:19-35), solveRam scalar branch (:61-71), prepare (:73-118),
// blend's lt/start arithmetic (:120-131).  numpy evaluates these in float64; so does this file, operation for
// operation, so that the tile grid is identical (tests/golden/planner.json).
#include "engine.h"

#include <algorithm>
#include <cmath>

namespace moe {

static int64_t ceil_by(int d, int64_t x) { return d <= 1 ? x : ((x + d - 1) / d) * d; }

AxisAnchors get_anchors(int64_t s, int64_t ns, int64_t l, int pad, int align, int sc)
{
    AxisAnchors a;
    const int64_t stride = l - 2 * pad;
    if (l >= ceil_by(align, s)) a.step = 1;
    else a.step = (int)std::max<int64_t>(2, (int64_t)std::ceil((double)ns / (double)stride));
    a.start.resize(a.step);
    a.end.resize(a.step);
    a.end_sc.resize(a.step);
    for (int k = 0; k < a.step; ++k) a.start[k] = k * stride + pad;
    a.start[0] = 0;
    for (int k = 0; k < a.step; ++k) { a.end[k] = a.start[k] + l; a.end_sc[k] = a.end[k] * sc; }
    if (a.step > 1) {
        a.start[a.step - 1] = s - ceil_by(align, s - a.end[a.step - 2] + pad);
        a.end[a.step - 1] = s;
        a.clip = (a.end[a.step - 2] - s) * sc;
    } else {
        a.end[0] = ceil_by(align, s);
        a.clip = 0;
    }
    a.end_sc[a.step - 1] = s * sc;
    return a;
}

// (first HR index written, first HR index taken un-blended, HR origin, HR extent of the tile result) per tile
static void axis_cover(const AxisAnchors& a, int sc, int pad_sc, int64_t out_len, std::vector<int>& tab,
                       std::vector<int>& first_of, std::vector<int>& cnt_of)
{
    tab.resize((size_t)a.step * 4);
    first_of.assign((size_t)out_len, 0);
    cnt_of.assign((size_t)out_len, 0);
    for (int i = 0; i < a.step; ++i) {
        const int64_t origin = a.start[i] * sc;
        const int64_t extent = (a.end[i] - a.start[i]) * sc;          // rows of the net's output for this tile
        const int64_t end = std::min<int64_t>(a.end_sc[i], out_len);
        const int64_t l = end - origin;                               // after unpad
        int64_t lt = (i == a.step - 1) ? a.clip : (i == 0 ? 0 : pad_sc);
        if (lt < 0) lt += l;
        int64_t first = origin, solid = origin;
        if (lt >= 1) { first = origin + lt - pad_sc; solid = origin + lt; }
        tab[i * 4 + 0] = (int)first;
        tab[i * 4 + 1] = (int)solid;
        tab[i * 4 + 2] = (int)origin;
        tab[i * 4 + 3] = (int)extent;
        for (int64_t y = first; y < end; ++y) {
            if (cnt_of[y] == 0) first_of[y] = i;
            cnt_of[y] += 1;
        }
    }
}

int build_plan(Plan& p, const int64_t shape[3], double ram, double ram_coef, int pad, int sc, int align, int cropsize,
               std::string& err)
{
    const int64_t c = shape[0], h = shape[1], w = shape[2];
    if (c < 1 || h < 1 || w < 1 || pad < 0 || sc < 1 || align < 1 || (align & (align - 1))) {
        err = "moe_plan_create: bad shape / pad / scale / align";
        return MOE_EINVAL;
    }
    p.C = (int)c; p.H = (int)h; p.W = (int)w; p.pad = pad; p.sc = sc; p.align = align;
    // solveRam: m / c * (ramCoef / shape[0]); fixChannel = 0 on this path so c == shape[0]
    const double n = ram / (double)c * (ram_coef / (double)c);
    const int64_t s = ceil_by(align, 28 + pad * 2);
    if (n < (double)(s * s)) {
        err = "Free memory space is " + std::to_string((long long)ram) + " bytes, which is not enough.";
        return MOE_ENOMEM;
    }
    const int64_t ph = std::max<int64_t>(1, h - pad * 3), pw = std::max<int64_t>(1, w - pad * 3);
    const int64_t lo = (int64_t)((double)s / align), hi = (int64_t)(n / (double)(align * s)) + 1;   // arange(lo, hi)
    double best = 0;
    bool have = false;
    // pass 1: minimal tile count; pass 2: among the minima, the candidate index closest to len/2
    const int64_t len = hi - lo;
    auto count = [&](int64_t k, int64_t& nsv, int64_t& msv) {
        const int64_t ns0 = lo + k;
        const int64_t ms0 = (int64_t)(n / (double)(align * align) / (double)ns0);
        nsv = ns0 * align; msv = ms0 * align;
        double nn = std::max(2.0, std::ceil((double)ph / (double)(nsv - 2 * pad)));
        double mn = std::max(2.0, std::ceil((double)pw / (double)(msv - 2 * pad)));
        if (nsv >= h) nn = 1;
        if (msv >= w) mn = 1;
        return nn * mn;
    };
    for (int64_t k = 0; k < len; ++k) {
        int64_t a, b;
        const double d = count(k, a, b);
        if (!have || d < best) { best = d; have = true; }
    }
    int64_t mina = -1;
    double bestdist = 0;
    for (int64_t k = 0; k < len; ++k) {
        int64_t a, b;
        if (count(k, a, b) == best) {
            const double dist = std::fabs((double)k - (double)len / 2.0);
            if (mina < 0 || dist < bestdist) { mina = k; bestdist = dist; }
        }
    }
    int64_t nsv, msv;
    count(mina, nsv, msv);
    const int64_t ah = ceil_by(align, h), aw = ceil_by(align, w), acs = ceil_by(align, cropsize);
    int64_t ih = nsv, iw = msv;
    if (cropsize > 0) { ih = std::min(acs, nsv); iw = std::min(acs, msv); }
    ih = std::min(ah, ih); iw = std::min(aw, iw);
    p.tile_h = (int)ih; p.tile_w = (int)iw;
    p.ah = get_anchors(h, ph, ih, pad, align, sc);
    p.aw = get_anchors(w, pw, iw, pad, align, sc);
    p.pad_sc = pad * sc; p.out_h = (int)(h * sc); p.out_w = (int)(w * sc);
    p.pad_h_to = p.pad_w_to = 0;
    if (p.ah.step > 1 && p.aw.step > 1) {}
    else if (p.ah.step > 1) p.pad_w_to = (int)aw;
    else if (p.aw.step > 1) p.pad_h_to = (int)ah;
    else { p.pad_w_to = (int)aw; p.pad_h_to = (int)ah; }
    p.tiles.clear();
    for (int i = 0; i < p.ah.step; ++i) {
        const int64_t tt = (i == p.ah.step - 1) ? p.ah.clip : (i == 0 ? 0 : p.pad_sc);
        for (int j = 0; j < p.aw.step; ++j) {
            const int64_t lt = (j == p.aw.step - 1) ? p.aw.clip : (j == 0 ? 0 : p.pad_sc);
            p.tiles.push_back({(int)p.ah.start[i], (int)p.ah.end[i], (int)p.aw.start[j], (int)p.aw.end[j], (int)tt, (int)lt,
                               (int)p.ah.end_sc[i], (int)p.aw.end_sc[j]});
        }
    }
    p.ramp.resize((size_t)p.pad_sc);
    for (int k = 0; k < p.pad_sc; ++k) {
        // ((arange(padSc, fp32) / padSc - .5) * 9).sigmoid(): the argument is formed in fp32 like torch does
        const float t = ((float)k / (float)p.pad_sc - 0.5f) * 9.0f;
        p.ramp[k] = (float)(1.0 / (1.0 + std::exp(-(double)t)));
    }
    axis_cover(p.ah, sc, p.pad_sc, p.out_h, p.row_tab, p.row_first, p.row_cnt);
    axis_cover(p.aw, sc, p.pad_sc, p.out_w, p.col_tab, p.col_first, p.col_cnt);
    // every output pixel must be covered (new_empty canvas is never read before written)
    for (int v : p.row_cnt) if (v < 1) { err = "planner: uncovered output row"; return MOE_EINVAL; }
    for (int v : p.col_cnt) if (v < 1) { err = "planner: uncovered output column"; return MOE_EINVAL; }
    // pool layout (tile k = C fp32 planes of its HR extent, raster order) and same-shape groups for batching
    p.tile_off.resize(p.tiles.size());
    size_t off = 0;
    p.groups.clear();
    for (size_t k = 0; k < p.tiles.size(); ++k) {
        const TileRect& t = p.tiles[k];
        const int th = t.bottom - t.top, tw = t.right - t.left;
        p.tile_off[k] = (long long)off;
        off += (size_t)c * (size_t)(th * sc) * (size_t)(tw * sc);
        size_t gi = 0;
        for (; gi < p.groups.size(); ++gi)
            if (p.groups[gi].th == th && p.groups[gi].tw == tw) break;
        if (gi == p.groups.size()) p.groups.push_back({th, tw, {}, 0});
        p.groups[gi].tiles.push_back((int)k);
    }
    p.pool_elems_per_plane_set = off;
    int slot = 0;
    for (auto& g : p.groups) { g.first_slot = slot; slot += (int)g.tiles.size(); }
    return MOE_OK;
}

}  // namespace moe
