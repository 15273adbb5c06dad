// engine.h -- host-side structures of the engine (not part of the C ABI).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/moephoto_amd.h"
#include "common.h"

namespace moe {

// ---- planner ---------------------------------------------------------------------------------------
struct AxisAnchors {
    std::vector<int64_t> start, end, end_sc;
    int64_t clip = 0;
    int step = 1;
};
AxisAnchors get_anchors(int64_t s, int64_t ns, int64_t l, int pad, int align, int sc);

struct TileRect { int top, bottom, left, right, top_t, left_t, bsc, rsc; };

struct PlanDeviceCache {       // device-side tables of a plan for one (layout, shard), built on first use
    int device = -1;
    int C = 0;
    int64_t sC = 0, sH = 0, sW = 0;
    int shard_index = 0, shard_count = 1;
    bool y_mult8 = false;      // every y_off entry is a multiple of 8 elements (16-byte aligned output planes)
    std::vector<int> group_first, group_count;   // per plan group: first slot / number of this shard's tiles
    void* blob = nullptr;      // one allocation: all tables below
    long long* x_off = nullptr;    // [ngroups-concatenated tiles][C]
    long long* y_off = nullptr;    // same order: offset of each plane inside the tile pool
    long long* tile_off = nullptr; // [n_tiles] raster order
    int *row_first = nullptr, *row_cnt = nullptr, *col_first = nullptr, *col_cnt = nullptr, *row_tab = nullptr, *col_tab = nullptr;
    float* ramp = nullptr;
    std::vector<std::pair<int, int*>> strip_tabs;   // moe_stitch_band: row tables in which one tile row is present as the strip of its blend band
};

struct FramesDeviceCache {     // offset tables of one multi-frame sharded run (moe_run_plan_tiles), cached per layout
    int device = -1, C = 0, n_frames = 0;
    int64_t sC = 0, sH = 0, sW = 0, frame_stride = 0;
    std::vector<long long> tile_dst;             // host copy of the (frame, tile) -> destination table the device tables were built from
    bool y_mult8 = false;
    std::vector<int> group_first, group_count;   // per plan group: first slot / number of owned (frame, tile) pairs
    void* blob = nullptr;
    long long* x_off = nullptr;
    long long* y_off = nullptr;
};

struct CustomOffsets { int device; std::vector<long long> host; long long* dev; };   // a caller-defined tile layout for moe_stitch

struct TileGroup { int th, tw; std::vector<int> tiles; int first_slot; };   // same-shaped tiles, slots in x_off order

struct Plan {
    int C = 0, H = 0, W = 0, pad = 0, sc = 1, align = 8;
    int tile_h = 0, tile_w = 0, pad_sc = 0, out_h = 0, out_w = 0, pad_h_to = 0, pad_w_to = 0;
    AxisAnchors ah, aw;
    std::vector<TileRect> tiles;
    std::vector<float> ramp;
    std::vector<int> row_tab, col_tab, row_first, row_cnt, col_first, col_cnt;
    std::vector<TileGroup> groups;
    std::vector<long long> tile_off;   // element offsets inside the pool
    size_t pool_elems_per_plane_set = 0;
    mutable std::vector<std::unique_ptr<PlanDeviceCache>> dev;   // a few entries at most (one per shard/layout seen)
    mutable std::vector<std::unique_ptr<FramesDeviceCache>> fdev;
    mutable std::vector<CustomOffsets> custom_off;
    mutable float* pool = nullptr;     // internal per-tile fp32 results (when the caller passes none)
    mutable size_t pool_elems = 0;
};
int build_plan(Plan& p, const int64_t shape[3], double ram, double ram_coef, int pad, int sc, int align, int cropsize,
               std::string& err);

}  // namespace moe
