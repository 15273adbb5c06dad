// engine.cpp -- host side of libmoephoto_amd.so: model objects, weight packing, the per-architecture kernel
// sequences, the device-resident doCrop (tile gather -> net -> stitch) and the extern "C" boundary.
//
// Reference for every sequence: python/models.py:108-223 (MyNet, Net2x/3x/4x, NetDN, SEDN/_Conv_Block),
// python/MoeNet_lite2.py:22-54 (Net), python/imageProcess.py:157-172 (doCrop).  See include/moephoto_amd.h.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace moe;

// =====================================================================================================
// errors
// =====================================================================================================
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? MOE_ENOMEM : MOE_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// =====================================================================================================
// model
// =====================================================================================================
namespace {

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool set = false;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

struct ConvLayer {               // one MFMA convolution
    int taps = 9, nseg = 1, nchunks = 1, r = 1;
    int cin = 64, cout = 64, k = 3;
    float slope = 1.f, scale = 1.f;
    bool per_plane = false;      // SEDN trans: weights rebuilt per plane by the SE kernel
    size_t w_hi = 0, w_lo = 0, bias = 0, bias_img = 0, w_plain = 0, bias_plain = 0, w_pk32 = 0;   // offsets into the device blob
    size_t w_arsb_lo = 0;        // ... and their low parts ((w - fp16(w)) * 2^11) in the same order, for conv64_x3.hip
    size_t wq_hi8 = 0, wq_lo8 = 0;   // ... w_hi 2^8 and w_lo 2^8 as fp8 e4m3 A fragments of v_mfma_scale_f32_32x32x64_f8f6f4 (conv64_q8.hip): [tap 9][half 2][lane 64][32 B]
    size_t w_arsb = 0;           // 3x3 64->64 trunk convs: A fragments of v_mfma_f32_16x16x32_f16 in conv64_x3.hip's order (the register-weight 16x16x32 form; round 2's arsb_fused.hip, retired, introduced it)
    size_t w_x3 = 0;             // 1x1, one segment, split precision: [chunk][w_lo | w_hi | w_hi] for the single-launch path (acc_mode 4)
    bool has_x3 = false;
    bool has_bias = false;
    int nfrag() const { return nseg * taps * 8; }
};

// an activation tensor: fp16 [B][H][W][C] and (split operands / the trunk stream of 'mixed') its low part, (v - fp16(v)) 2^11, as fp16 in the same layout --
// or, lo8, as the fp8 e4m3 word of that value / 4 (one byte a channel): the form conv64_q8.hip reads and writes between its own layers
struct Act { half_t* hi = nullptr; half_t* lo = nullptr; bool lo8 = false; };

struct Arena {                   // bump allocator over the net's workspace (dry run when base == nullptr)
    char* base = nullptr;
    size_t off = 0;
    void* take(size_t bytes)
    {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        return base ? base + a : nullptr;
    }
};

}  // namespace

// Kernel-form switches of one net.  Process-wide defaults come from the environment ONCE, when the net is created (MOE_* variables,
// kept for command-line A/B runs); after that only moe_net_set_option changes them -- the forward path reads no environment.
struct NetOptions {
    int conv_impl = 2;        // conv_impl   sp (2, default: the fast 3x3 kernels) | v1 (0: the generic kernel everywhere; debugging, not with 'mixed')
    int sp_impl = 1;          // sp_impl     auto (1: conv3x3_rw where it wins) | rw (2: conv3x3_rw for every epilogue it compiles) | sp (0: conv3x3_sp only)
    int tail_split = 1;       // tail_split  0 | r (1, default: the R branch's fused tail also splits its activation operand) | ru (2)
    int tail_form = 1;        // tail_form   sums (1, default: phase-class sums + aprons from conv3x3_rw, tapsum4) | planes (0: nine tap planes per phase, tapsum2)
    int up_fuse2 = 1;         // up_fuse2    1 (default): lite's last two upsampler stages + the folded tail in ONE launch (conv1x1_f2.hip, split operands) | 0: stage by stage (conv1x1.hip; same bits)
    int up_impl = 1;          // up_impl     ps4 (1, default: the fused-tail up-conv with all four phases in one workgroup, conv3x3_ps4.hip + tailadd) | rw (0: conv3x3_rw
                              //             per phase, phase-class sums, tapsum4 -- round 3's form, kept for A/B and for shapes ps4 does not take)
    bool conv1x1 = true;      // conv1x1     lite's 1x1 layers on conv1x1.hip (0: generic kernel)
    bool x3_fuse = true;      // x3_fuse     split-operand 3x3 64->64 layers as ONE launch (conv64_x3.hip; 0: three launches)
    bool arsb_fuse = true;    // arsb_fuse   single-pass ARSBs as one launch (0: two launches)
    int lo8 = 1;              // lo8         on (default): between conv64_q8 layers the low parts travel as fp8 words (64 instead of 128 bytes a pixel) | off
    int x3_impl = 0;          // x3_impl     auto (0, default: q8 for the SR nets, x3 for the DN nets -- see forward) | x3 (1: conv64_x3.hip, three fp16 products) |
                              //             q8 (2: conv64_q8.hip, the two correction products on fp8 operands)
    int k48 = 1;              // k48         1 (default): kernels that can skip the zero k-slice of the 48-channel nets do | 0: they run all four (A/B)
    int s64 = 1;              // s64         1 (default): SEDN's fused block tail on conv64_s.hip (streamed, per-plane weights in registers) | 0: conv3x3_sp<6>
    int branch_streams = 1;   // branch_streams  1 (default): small launch sets (the reference's own per-tile loop: 3 planes of <= 256 x 256 per forward) run the U branch on a second
                              //             HIP stream beside the trunk + R branch -- a launch of a few planes leaves CUs idle at its tail (702 ARSB patches over 256 workgroups) and the
                              //             other branch's workgroups take them; same kernels, same bits | 0: one stream
    int branch_groups = 0;    // branch_groups   persistent workgroups of the side stream's launches in that mode (0: 5/16 of the CUs for the x4 nets, 3/16 for x2 / x3 -- about the U
                              //             branch's share of the forward; a4: 29.0 / 27.4 / 27.9 / 28.4 ms per frame with 64 / 80 / 96 / 112, 34.1 with 48; a2: 20.1-20.2 with 24 .. 64,
                              //             21.6 on one stream -- profiles/r05/g_branch_streams.txt); the trunk + R branch launch max_groups minus that many.  A launch of 3 planes scales badly over 256
                              //             workgroups -- the per-tile loop takes 30.6 / 31.3 / 38.2 / 66.4 ms per frame with 256 / 192 / 128 / 64 (profiles/r05/f_small_launch_scaling.txt)
                              //             -- so the two branches are given disjoint shares of the chip instead of each launch spreading over all of it
    int exact_fuse = 1;       // exact_fuse  1 (default): an exact ARSB of a chain runs as ONE launch (arsb_sq.hip: conv_1's rows stay in LDS) | 0: conv_1, conv_2 on conv64_sq / conv64_q8
    int q8_impl = 1;          // q8_impl     s (1, default: conv64_sq.hip, the chain layers streamed down a column by an fp16 wave + an fp8 wave) | p (0: conv64_q8.hip, 8 x 32 patches)
                              // (the one-launch ARSB of the single-pass blocks is arsb32c.hip.  Earlier generations -- arsb_fused.hip, arsb32.hip (history at 689845f) and the streamed
                              // form arsb_s.hip (round 4: bit-identical, 7 % slower, both at the package power cap; history at 321d022, profiles/r04/d_arsb_streamed_vs_patch.txt) -- are
                              // no longer built)
    bool fuse_tail = true;    // fuse_tail   last upsampler conv + 64->1 / 48->1 tail conv in one kernel
    bool sedn_fuse = true;    // sedn_fuse   SEDN's fused block tail
    bool pool_fuse = true;    // pool_fuse   SE / FRM channel sums out of the producing conv's epilogue
    bool lite_lut = true;     // lite_lut    lite, fp16 inputs: the U branch as a table lookup inside the final sum (see moe_net::lut) | 0: computed
    bool stem2 = true;        // stem2       lite: conv_input2's output written by the stem in closed form (x times a fixed vector: StemArgs::w2), the 48 -> 48 1x1 conv not launched | 0: launched
    bool frm_pre = true;      // frm_pre     lite (fp16x3): the FRM gate of an LB from conv_2's INPUT (frm_pre_kernel), conv_2 stores gate * conv + x -- no frm_apply pass | 0: gate from conv_2's output, frm_apply
    int overlap_calls = 1;    // overlap_calls  1 (default): consecutive small forwards that the CALLER marks as independent of each other (moe_net_forward_ex with MOE_FWD_INPUT_SINCE_PREV:
                              //             "my input was complete when the previous forward of this net was enqueued" -- true of the reference's tile loop, whose inputs are slices of
                              //             ONE padded image, python/imageProcess.py:164-170) alternate between two internal (stream, workspace) sets: forward k+1 starts beside forward k
                              //             instead of behind it and the caller's blend; only its LAST kernel (the one that writes y) waits for the caller's stream | 0: every forward on
                              //             the caller's stream
    int overlap_fork = 0;     // overlap_fork   0 (default): such a forward does not fork its U branch onto a side stream as well -- two forwards in flight ARE the second stream | 1: it does
    int overlap_groups = 0;   // overlap_groups  persistent workgroups per launch of such a forward; 0 (default): half of the CUs -- two forwards in flight on half the chip each.
                              //             Measured on the reference-style loop of bench.py (profiles/r06/d_dropin_overlap_queues.txt; option off: 29.4 ms = 0.86 of the headline):
                              //             fork 0 / 128 groups 28.5 ms (0.89; 27.6 = 0.92 with moe_blend_tile), fork 1 / all groups 28.4 (0.89; 28.1), fork 0 / all groups 29.4,
                              //             fork 1 / 128 groups 35.8; GPU_MAX_HW_QUEUES = 8 or 16 instead of HIP's 4: 36-39 ms.  The chip is saturated either way: what separates the
                              //             loop from the device-resident path is the fixed cost of forty 3-plane launch sets (weight preloads, ramp-up and tail of every kernel), which
                              //             two forwards in flight hide only in part
    int calib_log = 0;        // calib_log   1: moe_net_calibrate prints every count's measured and predicted error to stderr (tools/calib_report.py)
    int auto_calibrate = 1;   // auto_calibrate  1 (default): moe_net_finalize(MOE_PREC_AUTO) measures the count of split-operand ARSBs on the loaded weights | 0: per-architecture defaults
    int exact_blocks_env = -1;   // MOE_EXACT_BLOCKS (moe_net_set_exact_blocks overrides)
    int tiles_per_batch = 0;  // tiles_per_batch   tiles of 256^2 pixels per launch set when the caller passes 0 (0: 32)
    int max_groups = 0;       // max_groups  persistent workgroups per launch (0: one per CU), applied at finalize
    int dbg = 0;              // dbg         timing-ablation bits of the conv kernels (results are wrong when set)
    std::string trace_key = "convt_R1.up1";
    bool arsb_trace = false;
    std::string repeat_key;   // repeat      "<layer key substring>:<n>": the bracketed launches (prof_begin sites) of matching layers are issued n times -- measurement only (tools/kernel_power.py:
    int repeat_n = 1;         //             one kernel looped by itself while rocm-smi samples the package power and clock); the launches are idempotent, results do not change

    static int tri(const char* v, const char* a0, const char* a1, const char* a2, int dflt)
    {
        if (!v) return dflt;
        if (a0 && !strcmp(v, a0)) return 0;
        if (a1 && !strcmp(v, a1)) return 1;
        if (a2 && !strcmp(v, a2)) return 2;
        return -1;
    }
    static int onoff(const char* v) { return !v ? -1 : (!strcmp(v, "0") || !strcmp(v, "off")) ? 0 : (!strcmp(v, "1") || !strcmp(v, "on")) ? 1 : -1; }
    // false: unknown key or value
    bool set(const std::string& key, const char* v)
    {
        if (!v) return false;
        auto flag = [&](bool& dst) { const int b = onoff(v); if (b < 0) return false; dst = b != 0; return true; };
        if (key == "conv_impl") { if (!strcmp(v, "sp")) conv_impl = 2; else if (!strcmp(v, "v1")) conv_impl = 0; else return false; return true; }
        if (key == "sp_impl") { const int t = tri(v, "sp", "auto", "rw", -1); if (t < 0) return false; sp_impl = t; return true; }
        if (key == "tail_split") { const int t = tri(v, "0", "r", "ru", -1); if (t < 0) return false; tail_split = t; return true; }
        if (key == "tail_form") { const int t = tri(v, "planes", "sums", nullptr, -1); if (t < 0) return false; tail_form = t; return true; }
        if (key == "up_impl") { const int t = tri(v, "rw", "ps4", nullptr, -1); if (t < 0) return false; up_impl = t; return true; }
        if (key == "up_fuse2") { const int t = onoff(v); if (t < 0) return false; up_fuse2 = t; return true; }
        if (key == "lo8") { const int t = onoff(v); if (t < 0) return false; lo8 = t; return true; }
        if (key == "x3_impl") { const int t = tri(v, "auto", "x3", "q8", -1); if (t < 0) return false; x3_impl = t; return true; }
        if (key == "k48") { const int t = onoff(v); if (t < 0) return false; k48 = t; return true; }
        if (key == "s64") { const int t = onoff(v); if (t < 0) return false; s64 = t; return true; }
        if (key == "auto_calibrate") { const int t = onoff(v); if (t < 0) return false; auto_calibrate = t; return true; }
        if (key == "overlap_calls") { const int t = onoff(v); if (t < 0) return false; overlap_calls = t; return true; }
        if (key == "overlap_fork") { const int t = onoff(v); if (t < 0) return false; overlap_fork = t; return true; }
        if (key == "overlap_groups") { const int t = atoi(v); if (t < 0) return false; overlap_groups = t; return true; }
        if (key == "calib_log") { const int t = onoff(v); if (t < 0) return false; calib_log = t; return true; }
        if (key == "branch_groups") { branch_groups = atoi(v); return branch_groups >= 0; }
        if (key == "branch_streams") { const int t = onoff(v); if (t < 0) return false; branch_streams = t; return true; }
        if (key == "exact_fuse") { const int t = onoff(v); if (t < 0) return false; exact_fuse = t; return true; }
        if (key == "q8_impl") { if (v && !strcmp(v, "s")) q8_impl = 1; else if (v && !strcmp(v, "p")) q8_impl = 0; else return false; return true; }
        if (key == "conv1x1") return flag(conv1x1);
        if (key == "x3_fuse") return flag(x3_fuse);
        if (key == "arsb_fuse") return flag(arsb_fuse);
        if (key == "fuse_tail") return flag(fuse_tail);
        if (key == "sedn_fuse") return flag(sedn_fuse);
        if (key == "pool_fuse") return flag(pool_fuse);
        if (key == "frm_pre") return flag(frm_pre);
        if (key == "stem2") return flag(stem2);
        if (key == "lite_lut") return flag(lite_lut);
        if (key == "dbg") { dbg = atoi(v); return true; }
        if (key == "tiles_per_batch") { tiles_per_batch = atoi(v); return tiles_per_batch >= 0; }
        if (key == "max_groups") { max_groups = atoi(v); return max_groups >= 0; }
        if (key == "trace_key") { trace_key = v; return true; }
        if (key == "repeat") {
            // parsed into locals first: a refused value ("up1:0", "k:-3", "nonsense") must leave the option as it was -- a stored count of zero would issue no launch at all
            // for the matching layers
            const char* c = strrchr(v, ':');
            if (!c) {
                if (*v && strcmp(v, "0")) return false;
                repeat_key.clear(); repeat_n = 1;
                return true;
            }
            const int cnt = atoi(c + 1);
            if (cnt < 1 || c == v) return false;
            repeat_key.assign(v, c - v); repeat_n = cnt;
            return true;
        }
        return false;
    }
    void from_env()
    {
        static const char* const names[][2] = {{"MOE_CONV_IMPL", "conv_impl"}, {"MOE_SP_IMPL", "sp_impl"}, {"MOE_TAIL_SPLIT", "tail_split"}, {"MOE_TAIL_FORM", "tail_form"}, {"MOE_UP_IMPL", "up_impl"}, {"MOE_UP_FUSE2", "up_fuse2"},
                                               {"MOE_CONV1X1", "conv1x1"}, {"MOE_Q8_IMPL", "q8_impl"}, {"MOE_EXACT_FUSE", "exact_fuse"}, {"MOE_BRANCH_STREAMS", "branch_streams"}, {"MOE_BRANCH_GROUPS", "branch_groups"}, {"MOE_AUTO_CALIBRATE", "auto_calibrate"}, {"MOE_OVERLAP_CALLS", "overlap_calls"}, {"MOE_OVERLAP_GROUPS", "overlap_groups"}, {"MOE_OVERLAP_FORK", "overlap_fork"}, {"MOE_S64", "s64"}, {"MOE_K48", "k48"}, {"MOE_X3_IMPL", "x3_impl"}, {"MOE_LO8", "lo8"}, {"MOE_X3_FUSE", "x3_fuse"}, {"MOE_ARSB_FUSE", "arsb_fuse"}, {"MOE_FUSE_TAIL", "fuse_tail"},
                                               {"MOE_SEDN_FUSE", "sedn_fuse"}, {"MOE_POOL_FUSE", "pool_fuse"}, {"MOE_FRM_PRE", "frm_pre"}, {"MOE_STEM2", "stem2"}, {"MOE_LITE_LUT", "lite_lut"}, {"MOE_DBG", "dbg"}, {"MOE_TRACE_KEY", "trace_key"},
                                               {"MOE_TILES_PER_BATCH", "tiles_per_batch"}, {"MOE_MAX_GROUPS", "max_groups"}};
        for (const auto& nv : names)
            if (const char* e = getenv(nv[0]))
                if (!set(nv[1], e)) fprintf(stderr, "moephoto_amd: %s=\"%s\" is not a value of option %s -- ignored\n", nv[0], e, nv[1]);      // (a typo in an A/B run must not pass silently)

        if (const char* e = getenv("MOE_EXACT_BLOCKS")) exact_blocks_env = atoi(e);
        arsb_trace = getenv("MOE_ARSB_TRACE") != nullptr;
    }
};

struct moe_net {
    int arch = 0, scale = 1;
    NetOptions opt;
    int C = 64;                  // real channel count (48 for NetDN / lite); tensors are padded to 64
    int stages = 1, r = 2;       // upsampler stages and their shuffle factor
    std::vector<Param> params;
    std::map<std::string, int> index;
    bool finalized = false;
    int device = -1, precision = MOE_PREC_FP16;
    int exact_blocks = -1;       // MOE_PREC_MIXED: leading ARSBs computed with split operands (-1: the calibrated count if there is one, else the per-architecture default)
    // calibration of THESE weights (moe_net_calibrate; run by moe_net_finalize(MOE_PREC_AUTO) on the ARSB nets): valid until a parameter changes
    bool calib_valid = false;
    // lite, fp16 inputs (round 6): the U branch (MoeNet_lite2.py:47,50: conv_input, uim, convt_I1) is POINTWISE -- 1x1 convs, pixel shuffles, PReLUs on a one-channel input -- so its
    // output at an HR pixel is a function of ONE input value and the pixel's phase: a table over the 65,536 fp16 bit patterns, filled once per checkpoint by the U branch's own
    // kernels run on an image of all patterns (bit-identical to computing it), [256 r][256 r] fp32.  lut_state: 0 not tried, 1 ready, -1 not available, 2 being built
    float* lut = nullptr; half_t* lut_in = nullptr; int lut_state = 0;
    int calib_blocks = -1;       // smallest count of split-operand ARSBs whose worst noise-tile error against the exact mode is within the target (-1: none is -> FP16X3)
    double calib_err = 0.0;      // that error
    int auto_resolved = -1;      // what MOE_PREC_AUTO resolved to at the last finalize with it (-1: not finalized that way since the parameters changed)
    // device weights
    char* blob = nullptr;
    size_t blob_bytes = 0;
    std::vector<ConvLayer> convs;
    std::map<std::string, int> conv_index;
    std::map<std::string, size_t> small;     // name -> blob offset of small fp32 / fp16 tables
    std::map<std::string, float> scalars;
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int max_groups = 256;
    // live kernel timing of selected conv layers (bench.py's roofline leg): hipEvent pairs on the launch stream
    std::vector<std::string> prof_keys;          // comma-separated substrings of moe_net_set_profile
    struct ProfRec { hipEvent_t e0 = nullptr, e1 = nullptr; int key = 0; double flops = 0; };
    std::vector<ProfRec> prof_ev;                // event pairs, reused across steps
    size_t prof_used = 0;
    // moe_net_forward's host offset tables: a ring of pinned host slots + device slots, copied asynchronously on the launch stream
    struct OffSlot { long long* host = nullptr; long long* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    OffSlot off_ring[4];
    int off_next = 0;
    // second stream of small launch sets (option branch_streams): the U branch forks behind the stem and joins in front of the branch sum
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // two (stream, workspace, side stream) sets for consecutive small forwards the caller declares independent (option overlap_calls, moe_net_forward_ex): a set's members are
    // swapped into ws / side / ev_* for the duration of its forward
    struct PipeSet { hipStream_t main = nullptr, side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, entry = nullptr, done = nullptr; char* ws = nullptr; size_t ws_bytes = 0; };
    PipeSet pipe[2];
    int pipe_next = 0;
    bool pipe_prev_valid = false;
    hipStream_t pipe_last_stream = nullptr;
    hipEvent_t out_gate = nullptr;       // set around such a forward: the kernel that writes the caller's y waits for this event (the caller's stream position at THIS call)
    // debug taps
    bool debug = false;
    struct Tap { float* dev = nullptr; int64_t shape[4] = {0, 0, 0, 0}; };
    std::map<std::string, Tap> taps;

    const Param* get(const std::string& n) const
    {
        auto it = index.find(n);
        return it == index.end() ? nullptr : &params[it->second];
    }
    void add(const std::string& n, std::vector<int64_t> shape)
    {
        index[n] = (int)params.size();
        Param p; p.name = n; p.shape = std::move(shape);
        params.push_back(std::move(p));
    }
};

static void declare_params(moe_net& n)
{
    const int C = n.C;
    auto arsb = [&](const std::string& p) {
        n.add(p + "conv_1.weight", {C, C, 3, 3});
        n.add(p + "relu.weight", {1});
        n.add(p + "conv_2.weight", {C, C, 3, 3});
        n.add(p + "scale.scale", {1});
    };
    if (n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X) {
        n.add("conv_input.weight", {64, 1, 3, 3});
        n.add("conv_input2.weight", {64, 64, 3, 3});
        n.add("relu.weight", {1});
        for (const char* br : {"u", "convt_R1"}) {
            for (int s = 0; s < n.stages; ++s) {
                const std::string p = std::string(br) + "." + std::to_string(s) + ".";
                n.add(p + "0.weight", {64 * n.r * n.r, 64, 3, 3});
                n.add(p + "0.bias", {64 * n.r * n.r});
                n.add(p + "2.weight", {1});
            }
            n.add(std::string(br) + "." + std::to_string(n.stages) + ".weight", {1, 64, 3, 3});
        }
        for (int i = 1; i <= 6; ++i) arsb("convt_F" + std::to_string(i) + ".0.");
    } else if (n.arch == MOE_ARCH_NETDN) {
        n.add("conv_input.weight", {48, 1, 3, 3});
        n.add("conv_input2.weight", {48, 48, 3, 3});
        n.add("relu.weight", {1});
        n.add("u.weight", {1, 48, 3, 3});
        n.add("convt_R1.weight", {1, 48, 3, 3});
        for (int i = 1; i <= 6; ++i) arsb("convt_F" + std::to_string(i) + ".0.");
    } else if (n.arch == MOE_ARCH_SEDN) {
        n.add("conv_input.weight", {64, 1, 3, 3});
        n.add("convt_R1.weight", {1, 64, 3, 3});
        for (int b = 0; b < 16; ++b) {
            const std::string p = "convt_F1." + std::to_string(b) + ".";
            n.add(p + "rblock.0.weight", {64, 64, 3, 3});
            n.add(p + "rblock.2.weight", {64, 64, 3, 3});
            n.add(p + "rblock.4.weight", {256, 64, 3, 3});
            n.add(p + "trans.0.weight", {64, 256, 1, 1});
            n.add(p + "conv_down.weight", {16, 256, 1, 1});
            n.add(p + "conv_up.weight", {256, 16, 1, 1});
        }
    } else {   // MOE_ARCH_LITE
        n.add("conv_input.weight", {48, 1, 1, 1});
        n.add("conv_input2.weight", {48, 48, 1, 1});
        n.add("relu.weight", {1});
        for (const char* br : {"ures", "uim"})
            for (int s = 0; s < n.stages; ++s) {
                const std::string p = std::string(br) + "." + std::to_string(s) + ".";
                n.add(p + "0.weight", {192, 48, 1, 1});
                n.add(p + "0.bias", {192});
                n.add(p + "2.weight", {1});
            }
        n.add("convt_R1.weight", {1, 48, 1, 1});
        n.add("convt_I1.weight", {1, 48, 1, 1});
        for (int k = 1; k <= 3; ++k) {
            const std::string p = "convt_F1" + std::to_string(k) + ".";
            n.add(p + "conv_1.weight", {48, 48, 3, 3});
            n.add(p + "conv_2.weight", {48, 48, 3, 3});
            n.add(p + "relu.weight", {1});
            n.add(p + "se.conv_du.0.weight", {3, 48, 1, 1});
            n.add(p + "se.conv_du.0.bias", {3});
            n.add(p + "se.conv_du.2.weight", {48, 3, 1, 1});
            n.add(p + "se.conv_du.2.bias", {48});
        }
    }
}

// =====================================================================================================
// weight packing
// =====================================================================================================
namespace {

struct BlobBuilder {
    std::vector<char> data;
    size_t take(size_t bytes)
    {
        const size_t a = (data.size() + 255) & ~(size_t)255;
        data.resize(a + bytes, 0);
        return a;
    }
    template <typename T> T* at(size_t off) { return (T*)(data.data() + off); }
};

// packed output channel n' = chunk*64 + cl  ->  original output channel, or -1 (padding).
// r > 1: chunk = sub-pixel (i*r + j), cl = channel c of the shuffled output: original = c*r*r + i*r + j
inline int orig_cout(int np, int cout, int r)
{
    const int chunk = np / 64, cl = np % 64;
    if (r > 1) {
        const int cs = cout / (r * r);
        return cl < cs ? cl * r * r + chunk : -1;
    }
    return np < cout ? np : -1;
}

// A-operand fragments of v_mfma_f32_32x32x16_f16 for D[cout][pixel]: fragment f = ((seg*taps + tap)*4 + ks)*2 + nblk,
// lane l holds W[cout = chunk*64 + nblk*32 + (l&31)][cin = seg*64 + ks*16 + 8*(l>>5) + e][tap], e = 0..7
void pack_conv(const Param& W, const Param* bias, int r, ConvLayer& L, BlobBuilder& bb, bool want_lo, bool want_plain, bool want_pk32,
               float fold = 1.f)   // ScaleLayer folded into the MFMA weights: conv(x, w) * s == conv(x, w * s), product formed in fp32
{
    const int cout = (int)W.shape[0], cin = (int)W.shape[1], k = (int)W.shape[2];
    L.cout = cout; L.cin = cin; L.k = k; L.r = r;
    L.taps = k * k;
    L.nseg = (cin + 63) / 64;
    L.nchunks = r > 1 ? r * r : (cout + 63) / 64;
    const int nfrag = L.nfrag();
    const size_t nel = (size_t)L.nchunks * nfrag * 512;
    L.w_hi = bb.take(nel * 2);
    if (want_lo) L.w_lo = bb.take(nel * 2);
    if (want_pk32) L.w_pk32 = bb.take(nel * 4);
    for (int chunk = 0; chunk < L.nchunks; ++chunk)
        for (int f = 0; f < nfrag; ++f) {
            const int nblk = f & 1, ks = (f >> 1) & 3, st = f >> 3;
            const int tap = st % L.taps, seg = st / L.taps;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int np = chunk * 64 + nblk * 32 + (l & 31);
                    const int ci = seg * 64 + ks * 16 + 8 * (l >> 5) + e;
                    const int oc = orig_cout(np, cout, r);
                    float v = 0.f;
                    if (oc >= 0 && ci < cin) v = W.data[((size_t)oc * cin + ci) * L.taps + tap] * fold;
                    const size_t idx = ((size_t)(chunk * nfrag + f) * 64 + l) * 8 + e;
                    const half_t hv = (half_t)v;
                    bb.at<half_t>(L.w_hi)[idx] = hv;
                    if (want_lo) bb.at<half_t>(L.w_lo)[idx] = (half_t)((v - (float)hv) * 2048.f);
                    if (want_pk32) bb.at<float>(L.w_pk32)[idx] = v;
                }
        }
    if (want_lo && L.taps == 1 && L.nseg == 1) {
        // single-launch split precision (conv_mfma_kernel<1,3>): per chunk the 8 fragments of w_lo, then those of w_hi twice
        L.has_x3 = true;
        L.w_x3 = bb.take(nel * 2 * 3);
        const size_t per = (size_t)nfrag * 512;               // elements of one chunk
        for (int chunk = 0; chunk < L.nchunks; ++chunk)
            for (int seg3 = 0; seg3 < 3; ++seg3)
                memcpy(bb.at<half_t>(L.w_x3) + ((size_t)chunk * 3 + seg3) * per,
                       bb.at<half_t>(seg3 == 0 ? L.w_lo : L.w_hi) + (size_t)chunk * per, per * 2);
    }
    L.has_bias = bias != nullptr;
    L.bias_img = bb.take((size_t)L.nchunks * 256 * 4);        // zero-filled; the biases are written below
    if (bias) {
        L.bias = bb.take((size_t)L.nchunks * 64 * 4);
        for (int np = 0; np < L.nchunks * 64; ++np) {
            const int oc = orig_cout(np, cout, r);
            bb.at<float>(L.bias)[np] = oc >= 0 ? bias->data[oc] * fold : 0.f;
            bb.at<float>(L.bias_img)[(np / 64) * 256 + (np % 64)] = bb.at<float>(L.bias)[np];
        }
    }
    if (want_plain) {
        L.w_plain = bb.take(W.data.size() * 4);
        memcpy(bb.at<float>(L.w_plain), W.data.data(), W.data.size() * 4);
        if (bias) {
            L.bias_plain = bb.take(bias->data.size() * 4);
            memcpy(bb.at<float>(L.bias_plain), bias->data.data(), bias->data.size() * 4);
        }
    }
}

}  // namespace

// conv64_x3.hip's weight order (introduced by round 2's arsb_fused.hip, retired): [wave w][fragment f = tap * 2 + kh][lane l][e], lane l = (m = l & 15, kq = l >> 4) holds
// W[cout = 16w + m][cin = 8 * SL(kh, kq) + e][tap] with SL(kh, kq) = (2kh + (kq >> 1)) ^ 4(kq & 1) -- the k order in which that kernel's
// bank-conflict-free LDS image delivers the activations
static void pack_arsb(const Param& W, ConvLayer& L, BlobBuilder& bb, float fold, bool want_lo)
{
    const int cout = (int)W.shape[0], cin = (int)W.shape[1];
    L.w_arsb = bb.take((size_t)4 * 18 * 512 * 2);
    if (want_lo) L.w_arsb_lo = bb.take((size_t)4 * 18 * 512 * 2);
    for (int w = 0; w < 4; ++w)
        for (int f = 0; f < 18; ++f) {
            const int tap = f >> 1, kh = f & 1;
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, kq = l >> 4;
                const int slot = (2 * kh + (kq >> 1)) ^ (4 * (kq & 1));
                for (int e = 0; e < 8; ++e) {
                    const int oc = 16 * w + m, ci = 8 * slot + e;
                    const float v = (oc < cout && ci < cin) ? W.data[((size_t)oc * cin + ci) * 9 + tap] * fold : 0.f;
                    const half_t hv = (half_t)v;
                    bb.at<half_t>(L.w_arsb)[((size_t)(w * 18 + f) * 64 + l) * 8 + e] = hv;
                    if (want_lo) bb.at<half_t>(L.w_arsb_lo)[((size_t)(w * 18 + f) * 64 + l) * 8 + e] = (half_t)((v - (float)hv) * 2048.f);
                }
            }
        }
}

// OCP fp8 e4m3 (4 exponent bits, bias 7, no infinities, one NaN code; largest finite value 448), round to nearest even, saturating
static unsigned char to_e4m3(float v)
{
    const unsigned char sign = std::signbit(v) ? 0x80 : 0;
    float x = std::fabs(v);
    if (!(x == x)) return sign | 0x7F;
    if (x >= 448.f) return sign | 0x7E;
    if (x < 0.0009765625f) return sign;                       // below half of the smallest subnormal (2^-9): zero (ties to even: 2^-10 -> 0)
    int e;
    (void)std::frexp(x, &e);                                  // x = m 2^e, m in [0.5, 1)
    int ex = e - 1;                                           // x = 1.f 2^ex
    if (ex < -6) ex = -6;                                     // subnormal: spacing 2^-9
    const float q = std::ldexp(1.f, ex - 3);                  // spacing of the grid at this exponent
    float n = std::nearbyint(x / q);                          // (default rounding mode: to nearest even)
    float y = n * q;
    if (y >= 448.f) return sign | 0x7E;
    if (y < 0.015625f) return sign | (unsigned char)(int)std::nearbyint(y * 512.f);      // subnormal: mantissa = y / 2^-9
    int e2;
    const float m2 = std::frexp(y, &e2);                      // y = m2 2^e2
    const int ebits = e2 - 1 + 7;
    const int mbits = (int)std::nearbyint((m2 * 2.f - 1.f) * 8.f);
    return sign | (unsigned char)((ebits << 3) | mbits);
}

// fp8 weight parts of a 3x3 64->64 conv for conv64_q8.hip: lane l of fragment (tap, half c) holds output channel 32c + (l & 31), input channels
// 32 (l >> 5) .. + 31 of that tap; both parts carry a factor 2^8 (the kernel's E8M0 scale takes it out again)
static void pack_q8(const Param& W, ConvLayer& L, BlobBuilder& bb, float fold)
{
    const int cout = (int)W.shape[0], cin = (int)W.shape[1];
    // a weight of 1.75 or more would saturate (448 / 2^8) and its correction product would be wrong by the excess: such a layer keeps conv64_x3
    for (size_t i = 0; i < (size_t)cout * cin * 9; ++i)
        if (!(std::fabs(W.data[i] * fold) < 1.75f)) return;
    L.wq_hi8 = bb.take((size_t)9 * 2 * 64 * 32);
    L.wq_lo8 = bb.take((size_t)9 * 2 * 64 * 32);
    for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < 2; ++c)
            for (int l = 0; l < 64; ++l)
                for (int b = 0; b < 32; ++b) {
                    const int oc = 32 * c + (l & 31), ci = 32 * (l >> 5) + b;
                    const float v = (oc < cout && ci < cin) ? W.data[((size_t)oc * cin + ci) * 9 + tap] * fold : 0.f;
                    const half_t hv = (half_t)v;
                    const size_t idx = ((size_t)(tap * 2 + c) * 64 + l) * 32 + b;
                    bb.at<unsigned char>(L.wq_hi8)[idx] = to_e4m3((float)hv * 256.f);
                    bb.at<unsigned char>(L.wq_lo8)[idx] = to_e4m3((float)(half_t)((v - (float)hv) * 2048.f) * 256.f);
                }
}

static float scalar_of(const moe_net& n, const std::string& name) { return n.get(name)->data[0]; }

static void drop_lut(moe_net& n)
{
    if (n.lut || n.lut_in) (void)hipDeviceSynchronize();      // (a forward reading the table may be in flight; rare: weights or options changed)
    if (n.lut) (void)hipFree(n.lut);
    if (n.lut_in) (void)hipFree(n.lut_in);
    n.lut = nullptr; n.lut_in = nullptr; n.lut_state = 0;
}

static int build_device_weights(moe_net& n, int precision)
{
    drop_lut(n);      // (the table holds the U branch of the weights packed before)

    BlobBuilder bb;
    n.convs.clear(); n.conv_index.clear(); n.small.clear(); n.scalars.clear();
    const bool mixed = precision == MOE_PREC_MIXED, plain = precision == MOE_PREC_DEBUG_DIRECT;
    auto conv = [&](const std::string& key, const std::string& wname, const char* bname, int r, float slope, float scale,
                    bool per_plane = false) {
        ConvLayer L;
        const Param* b = bname ? n.get(bname) : nullptr;
        // low-order weight parts: every conv under FP16X3; under MIXED the trunk convs that may run with split operands
        const bool lo = precision == MOE_PREC_FP16X3 || (mixed && (key == "input2" || key.compare(0, 3, "c1_") == 0 || key.compare(0, 3, "c2_") == 0));
        // the debug path keeps the plain weights and applies the scale in its epilogue; the MFMA kernels get it pre-multiplied
        pack_conv(*n.get(wname), b, r, L, bb, lo, plain, per_plane, plain ? 1.f : scale);
        L.slope = slope; L.scale = plain ? scale : 1.f; L.per_plane = per_plane;
        // every bias-free 3x3 conv with <= 64 channels in and out (the trunks of Net*x / NetDN, lite's LB convs, SEDN's rblock.0/2) also gets
        // the register-resident weight order of conv64_x3.hip (+ its low part where split operands may be asked for)
        if (!plain && r == 1 && L.taps == 9 && L.nseg == 1 && L.nchunks == 1 && !b && !per_plane)
        {
            pack_arsb(*n.get(wname), L, bb, scale, lo);
            if (lo) pack_q8(*n.get(wname), L, bb, scale);
        }
        n.conv_index[key] = (int)n.convs.size();
        n.convs.push_back(L);
    };
    auto f32table = [&](const std::string& key, const std::vector<float>& v) {
        const size_t o = bb.take(v.size() * 4);
        memcpy(bb.at<float>(o), v.data(), v.size() * 4);
        n.small[key] = o;
    };
    auto stem = [&](const std::string& wname) {   // [tap][64] fp32
        const Param& W = *n.get(wname);
        const int C = (int)W.shape[0], taps = (int)(W.shape[2] * W.shape[3]);
        std::vector<float> t((size_t)taps * 64, 0.f);
        for (int c = 0; c < C; ++c) for (int k = 0; k < taps; ++k) t[(size_t)k * 64 + c] = W.data[(size_t)c * taps + k];
        f32table("stem", t);
        n.scalars["stem_taps"] = (float)taps;
    };
    auto tail = [&](const std::string& key, const std::string& wname) {   // [tap][64] fp16 (+lo)
        const Param& W = *n.get(wname);
        const int C = (int)W.shape[1], taps = (int)(W.shape[2] * W.shape[3]);
        const size_t o = bb.take((size_t)taps * 64 * 2), ol = bb.take((size_t)taps * 64 * 2);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < taps; ++k) {
                const float v = W.data[(size_t)c * taps + k];
                const half_t hv = (half_t)v;
                bb.at<half_t>(o)[(size_t)k * 64 + c] = hv;
                bb.at<half_t>(ol)[(size_t)k * 64 + c] = (half_t)((v - (float)hv) * 2048.f);
            }
        n.small[key] = o; n.small[key + ".lo"] = ol;
        n.scalars["tail_taps"] = (float)taps;
        if (taps == 9) {   // A fragments for the fused tail (conv3x3_sp EPI 3): slice i = 16 channels, lane row = tap, k = (hh, e)
            // Rows 0..8 carry the fp16 weights of the nine taps, rows 16..24 their rounding remainders (w - fp16(w)) * 2^11: the same
            // MFMA that forms the tap sums also forms the low-order sums (23 of the 32 rows were idle), and the epilogue adds
            // row 16+t * 2^-11 to row t -- the tail conv sees its weights to ~22 bits at no extra matrix work.
            // Fragments 4..7 (EPI 7, the activation operand split as well): the fp16 weights again, but in rows 16..24 -- multiplied by the
            // activations' low parts (units of 2^-11) they accumulate into the same low-order rows.
            const size_t of = bb.take(8 * 512 * 2);
            for (int i = 0; i < 8; ++i)
                for (int l = 0; l < 64; ++l)
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int row = l & 31, hh = l >> 5;
                        const int tap = row < 9 ? row : (row >= 16 && row < 25 ? row - 16 : -1);
                        const int ch = (i & 3) * 16 + (e8 < 4 ? 4 * hh + e8 : 8 + 4 * hh + (e8 - 4));
                        float v = 0.f;
                        if (tap >= 0 && ch < C) {
                            const float wv = W.data[(size_t)ch * taps + tap];
                            const half_t hv = (half_t)wv;
                            if (i < 4) v = row < 9 ? (float)hv : (wv - (float)hv) * 2048.f;
                            else v = row < 9 ? 0.f : (float)hv;
                        }
                        bb.at<half_t>(of)[(i * 64 + l) * 8 + e8] = (half_t)v;
                    }
            n.small[key + ".frag"] = of;
        }
    };
    n.small["zero"] = bb.take(1024);
    n.small["trash"] = bb.take(4096);
    n.small["zero_bias"] = bb.take(1024 * 4);      // up to 16 chunks of 64 fp32 zeros
    n.small["zero_bias_img"] = bb.take(16 * 256 * 4);   // the same as 1-KiB-per-chunk images (conv3x3_sp)

    if (n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X || n.arch == MOE_ARCH_NETDN) {
        stem("conv_input.weight");
        n.scalars["stem_slope"] = scalar_of(n, "relu.weight");
        conv("input2", "conv_input2.weight", nullptr, 1, 1.f, 1.f);
        for (int i = 1; i <= 6; ++i) {
            const std::string p = "convt_F" + std::to_string(i) + ".0.";
            conv("c1_" + std::to_string(i), p + "conv_1.weight", nullptr, 1, scalar_of(n, p + "relu.weight"), 1.f);
            conv("c2_" + std::to_string(i), p + "conv_2.weight", nullptr, 1, 1.f, scalar_of(n, p + "scale.scale"));
        }
        if (n.arch == MOE_ARCH_NETDN) {
            tail("tail_r", "convt_R1.weight");
            tail("tail_u", "u.weight");
        } else {
            for (const char* br : {"u", "convt_R1"}) {
                for (int s = 0; s < n.stages; ++s) {
                    const std::string p = std::string(br) + "." + std::to_string(s) + ".";
                    conv(std::string(br) + ".up" + std::to_string(s), p + "0.weight", (p + "0.bias").c_str(), n.r,
                         scalar_of(n, p + "2.weight"), 1.f);
                }
            }
            tail("tail_r", "convt_R1." + std::to_string(n.stages) + ".weight");
            tail("tail_u", "u." + std::to_string(n.stages) + ".weight");
        }
    } else if (n.arch == MOE_ARCH_SEDN) {
        stem("conv_input.weight");
        n.scalars["stem_slope"] = 0.2f;
        for (int b = 0; b < 16; ++b) {
            const std::string p = "convt_F1." + std::to_string(b) + ".", k = "b" + std::to_string(b);
            conv(k + ".rb0", p + "rblock.0.weight", nullptr, 1, 0.2f, 1.f);
            conv(k + ".rb2", p + "rblock.2.weight", nullptr, 1, 0.2f, 1.f);
            conv(k + ".rb4", p + "rblock.4.weight", nullptr, 1, 1.f, 1.f);
            conv(k + ".trans", p + "trans.0.weight", nullptr, 1, 0.2f, 1.f, true);
            f32table(k + ".down", n.get(p + "conv_down.weight")->data);
            f32table(k + ".up", n.get(p + "conv_up.weight")->data);
            {   // fused block tail (sedn_fuse): rblock.4 as [256][tap*64 + ci] and transposed, trans as [64][256], all fp32
                const Param& W4 = *n.get(p + "rblock.4.weight");
                std::vector<float> w256((size_t)256 * 576), w256t((size_t)576 * 256);
                for (int m = 0; m < 256; ++m)
                    for (int ci = 0; ci < 64; ++ci)
                        for (int tap = 0; tap < 9; ++tap) {
                            const float v = W4.data[((size_t)m * 64 + ci) * 9 + tap];
                            w256[(size_t)m * 576 + tap * 64 + ci] = v;
                            w256t[(size_t)(tap * 64 + ci) * 256 + m] = v;
                        }
                f32table(k + ".w256", w256);
                f32table(k + ".w256t", w256t);
                f32table(k + ".wt", n.get(p + "trans.0.weight")->data);
            }
        }
        tail("tail_r", "convt_R1.weight");
    } else {
        stem("conv_input.weight");
        n.scalars["stem_slope"] = scalar_of(n, "relu.weight");
        conv("input2", "conv_input2.weight", nullptr, 1, 1.f, 1.f);
        {   // conv_input2(PReLU(conv_input(x))) in closed form (MoeNet_lite2.py:40-41: one input channel, 1x1 kernels, no bias): PReLU(w_s[c] x) = x p[c] for x >= 0 and
            // x q[c] for x < 0 (p, q = w_s[c] or slope w_s[c] by the sign of w_s[c]), so conv_input2's output is x P / x Q with P = W2 p, Q = W2 q, summed in double here
            const Param& Ws = *n.get("conv_input.weight");
            const Param& W2 = *n.get("conv_input2.weight");
            const double sl = (double)scalar_of(n, "relu.weight");
            std::vector<float> pq(2 * 64, 0.f);
            for (int co = 0; co < 48; ++co) {
                double P = 0.0, Q = 0.0;
                for (int c = 0; c < 48; ++c) {
                    const double w = (double)Ws.data[c], w2 = (double)W2.data[(size_t)co * 48 + c];
                    P += w2 * (w >= 0.0 ? w : sl * w);
                    Q += w2 * (w <= 0.0 ? w : sl * w);
                }
                pq[co] = (float)P; pq[64 + co] = (float)Q;
            }
            f32table("stem.p2", pq);
        }
        for (int k = 1; k <= 3; ++k) {
            const std::string p = "convt_F1" + std::to_string(k) + ".", key = "lb" + std::to_string(k);
            conv(key + ".c1", p + "conv_1.weight", nullptr, 1, scalar_of(n, p + "relu.weight"), 1.f);
            conv(key + ".c2", p + "conv_2.weight", nullptr, 1, 1.f, 1.f);
            // FRM tables padded to 64 channels: w0 [3][64], b0 [3], w2 [64][3], b2 [64]
            std::vector<float> w0(3 * 64, 0.f), w2(64 * 3, 0.f), b2(64, 0.f);
            const Param& W0 = *n.get(p + "se.conv_du.0.weight");
            const Param& W2 = *n.get(p + "se.conv_du.2.weight");
            for (int j = 0; j < 3; ++j) for (int c = 0; c < 48; ++c) w0[j * 64 + c] = W0.data[j * 48 + c];
            for (int c = 0; c < 48; ++c) { for (int j = 0; j < 3; ++j) w2[c * 3 + j] = W2.data[c * 3 + j]; b2[c] = n.get(p + "se.conv_du.2.bias")->data[c]; }
            f32table(key + ".w0", w0);
            f32table(key + ".b0", n.get(p + "se.conv_du.0.bias")->data);
            f32table(key + ".w2", w2);
            f32table(key + ".b2", b2);
            // conv_2's weights once more, fp32 and transposed: [tap*64 + ci][co 64] -- the gate's pooled mean is this matrix applied to the window sums of conv_2's input (frm_pre)
            std::vector<float> c2t((size_t)576 * 64, 0.f);
            const Param& C2 = *n.get(p + "conv_2.weight");
            for (int co = 0; co < 48; ++co)
                for (int ci = 0; ci < 48; ++ci)
                    for (int tap = 0; tap < 9; ++tap) c2t[(size_t)(tap * 64 + ci) * 64 + co] = C2.data[((size_t)co * 48 + ci) * 9 + tap];
            f32table(key + ".c2t", c2t);
        }
        for (const char* br : {"ures", "uim"})
            for (int s = 0; s < n.stages; ++s) {
                const std::string p = std::string(br) + "." + std::to_string(s) + ".";
                conv(std::string(br) + ".up" + std::to_string(s), p + "0.weight", (p + "0.bias").c_str(), 2, scalar_of(n, p + "2.weight"), 1.f);
            }
        tail("tail_r", "convt_R1.weight");
        tail("tail_u", "convt_I1.weight");
        for (const char* kv : {"tail_r", "tail_u"}) {     // fp32 copies for the fused 1x1 tail (conv_mfma_kernel, tail1_w)
            const Param& W = *n.get(std::string(kv) == "tail_r" ? "convt_R1.weight" : "convt_I1.weight");
            std::vector<float> t(64, 0.f);
            for (int c = 0; c < (int)W.shape[1]; ++c) t[c] = W.data[c];
            f32table(std::string(kv) + ".f32", t);
        }
    }

    if (n.blob) { (void)hipFree(n.blob); n.blob = nullptr; }
    HIP_TRY(hipMalloc((void**)&n.blob, bb.data.size()));
    n.blob_bytes = bb.data.size();
    HIP_TRY(hipMemcpy(n.blob, bb.data.data(), bb.data.size(), hipMemcpyHostToDevice));
    return MOE_OK;
}

// =====================================================================================================
// forward
// =====================================================================================================
namespace {

struct Fwd {
    moe_net& n;
    hipStream_t s;
    int B, h, w;
    Arena ar;
    bool x3, direct;
    bool mixed = false;          // MOE_PREC_MIXED: fp16 operands, fp32-equivalent (hi + lo) trunk stream, split operands on selected layers
    bool y_vec = false;
    float* acc32 = nullptr;
    size_t acc32_elems = 0;
    half_t* side16 = nullptr;    // fp16 sum of the two low-order products of a 3x3 conv (split precision), output layout
    bool dry() const { return ar.base == nullptr; }
    float* lut_capture = nullptr;     // lite: this forward fills the U-branch table -- its input is the image of all fp16 patterns; part[1] is copied here instead of summed
    int tail1_parts = 2;         // partial planes per branch the fused 1x1 tail wrote (conv_mfma_kernel: 2, conv1x1.hip: 1)
    int tail_form = 0;           // fused tail of this forward: 0 nine tap planes (conv3x3_sp), 1 phase-class sums (conv3x3_rw + tapsum4)
    float* pool_out = nullptr;   // set around a conv() call: let the conv pool its output per plane (conv64_x3's pooled epilogue), [B][pool_slabs][64]
    int pool_slabs = 0;
    bool pool_done = false;      // the conv did
    bool pool_act = false;       // ... with pool_out: the conv may pool BEHIND its PReLU (conv64_x3 EPI 4: lite's conv_1, whose output's sums make the FRM gate -- frm_pre)
    const float* gate_in = nullptr;   // set around a conv() call with a residual: out = gate[plane][channel] * conv + residual (conv64_x3 EPI 5), [2][B][64]
    bool gate_done = false;      // the conv did

    Act act(long long pixels, int ch = 64, bool want_lo = false)
    {
        Act a;
        // + 2 KiB slack: the branch-free conv epilogue parks its predicated-off lanes just behind the last element
        a.hi = (half_t*)ar.take((size_t)pixels * ch * 2 + 2048);
        if (x3 || want_lo) a.lo = (half_t*)ar.take((size_t)pixels * ch * 2 + 2048);
        return a;
    }
    template <typename T> T* blob(size_t off) const { return (T*)(n.blob + off); }
    template <typename T> T* small(const std::string& k) const { return (T*)(n.blob + n.small.at(k)); }

    void tap(const std::string& name, const Act& a, int H, int W, int cs, int C)
    {
        if (!n.debug || dry()) return;
        auto& t = n.taps[name];
        if (t.dev) { (void)hipFree(t.dev); t.dev = nullptr; }
        const size_t nel = (size_t)B * C * H * W;
        if (hipMalloc((void**)&t.dev, nel * 4) != hipSuccess) return;
        t.shape[0] = B; t.shape[1] = C; t.shape[2] = H; t.shape[3] = W;
        launch_nhwc_to_nchw_f32(a.hi, a.lo, t.dev, B, H, W, cs, C, s);
    }

    // MIXED: which fused-tail launches also split the activation operand.  The R branch (trunk -> upsampler -> tail) carries the larger
    // share of the remaining error (emulation: r.tail activations 3.8e-4 vs u.tail 1.3e-4 on noise); MOE_TAIL_SPLIT = 0 | r (default) | ru
    bool tail_split_for(const std::string& key) const
    {
        const int mode = n.opt.tail_split;
        return mode == 2 || (mode == 1 && key.compare(0, 8, "convt_R1") == 0);
    }

    // live timing (bench.py's roofline legs): a hipEvent pair on the launch stream around the launches of a layer whose key matches
    // one of the profile substrings.  Returns the record index or -1.
    int prof_begin(const std::string& key, double flops)
    {
        for (size_t i = 0; i < n.prof_keys.size(); ++i) {
            if (key.find(n.prof_keys[i]) == std::string::npos) continue;
            if (n.prof_used == n.prof_ev.size()) {
                moe_net::ProfRec r;
                if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
                n.prof_ev.push_back(r);
            }
            moe_net::ProfRec& r = n.prof_ev[n.prof_used];
            r.key = (int)i; r.flops = flops;
            (void)hipEventRecord(r.e0, s);
            return (int)n.prof_used++;
        }
        return -1;
    }
    void prof_end(int rec) { if (rec >= 0) (void)hipEventRecord(n.prof_ev[rec].e1, s); }
    int repeats(const std::string& key) const { return (!n.opt.repeat_key.empty() && key.find(n.opt.repeat_key) != std::string::npos) ? n.opt.repeat_n : 1; }

    // one convolution layer: in [B][H][W][64*nseg] -> out [B][H*r][W*r][r>1 ? 64 : 64*nchunks]
    // returns false only when asked for the fused tail (tplanes != nullptr) and the fused kernel cannot take the layer
    // The two correction products on fp8 operands (conv64_q8.hip): 'mixed' only -- 'fp16x3' promises 2e-5, fp8 corrections deliver ~15 bits.
    // auto: the SR nets, whose all-tile sweep keeps its margin with it (worst 8.1e-4 either way, profiles/r03/m_conv64_q8.txt); the DN nets
    // (dn_lite5 7.2e-4 -> 8.6e-4 of the 1e-3 bar) stay on three fp16 products.
    bool use_q8() const { return mixed && (n.opt.x3_impl == 2 || (n.opt.x3_impl == 0 && (n.scale > 1 || n.arch == MOE_ARCH_NETDN))); }      // (round 6: NetDN too -- dn_lite5 7.6e-4 against 8.1e-4 on conv64_x3, and faster)
    // what conv() asks of a layer before it hands it to conv64_q8 (besides the tensors' own conditions)
    bool q8_capable(const ConvLayer& L) const
    {
        return n.opt.x3_fuse && n.opt.conv_impl == 2 && L.k == 3 && L.r == 1 && L.nchunks == 1 && L.nseg == 1 && !L.per_plane && L.wq_hi8 && L.w_arsb_lo && !L.has_bias && L.slope <= 1.f &&
               (long long)B * Hq * Wq * 128 + (Wq + 1ll) * 128 < (1ll << 32) - 65536;
    }
    int Hq = 0, Wq = 0;      // the trunk's resolution (set by forward before it plans the chain)

    bool conv(const std::string& key, const Act& in, const Act& out, const Act* res, int H, int W, const half_t* plane_w = nullptr,
              const half_t* plane_w_lo = nullptr, const half_t* tail_w = nullptr, float* tplanes = nullptr,
              const float* tail1_w = nullptr, float* tail1_out = nullptr, bool exact = false)
    {
        if (dry()) return true;
        const bool x3 = this->x3 || exact;      // split operands for this layer (every layer under FP16X3, selected ones under MIXED)
        const ConvLayer& L = n.convs[n.conv_index.at(key)];
        const bool any8 = in.lo8 || out.lo8 || (res && res->lo8);      // fp8 low parts: conv64_q8 or nothing (the caller planned the chain with q8_chain_ok)
        if (any8 && !(x3 && use_q8() && q8_capable(L) && !direct)) return false;
        const int out_cs = L.r > 1 ? 64 : 64 * L.nchunks;
        if (direct) {
            DirectConvArgs d{};
            d.in = in.hi; d.out = out.hi; d.res = res ? res->hi : nullptr;
            d.w = blob<float>(L.w_plain);
            d.bias = L.has_bias ? blob<float>(L.bias_plain) : nullptr;
            d.w_batch_stride = 0;
            d.B = B; d.H = H; d.W = W; d.in_cs = 64 * L.nseg; d.out_cs = out_cs; d.cin = L.cin; d.cout = L.cout; d.k = L.k; d.r = L.r;
            d.slope = L.slope; d.scale = L.scale;
            if (L.per_plane) { d.w = (const float*)plane_w; d.w_batch_stride = (long long)L.cout * L.cin; }
            launch_conv_direct(d, s);
            return true;
        }
        ConvArgs a{};
        a.in = in.hi; a.out = out.hi; a.res = res ? res->hi : nullptr;
        a.wpk = L.per_plane ? plane_w : blob<half_t>(L.w_hi);
        a.bias = L.has_bias ? blob<float>(L.bias) : nullptr;
        a.zero = small<half_t>("zero");
        a.bias_img = blob<float>(L.bias_img);
        a.trash = small<half_t>("trash");
        if (!a.bias) a.bias = small<float>("zero_bias");   // kernels initialise their accumulators from the bias vector
        a.w_batch_stride = L.per_plane ? (long long)L.nchunks * L.nfrag() * 512 : 0;
        a.B = B; a.H = H; a.W = W; a.in_cs = 64 * L.nseg; a.out_cs = out_cs; a.r = L.r; a.nchunks = L.nchunks;
        a.px = (W + kTileW - 1) / kTileW; a.py = (H + kTileH - 1) / kTileH;
        const long long items = (long long)B * a.px * a.py;
        // Workgroup (chunk, g) is block ((g / 8) * nchunks + chunk) * 8 + g % 8 and blocks go round-robin to the 8 XCDs, so an XCD gets
        // nchunks * ceil(G / 8) persistent workgroups: keep that within its CUs (one 160-KiB workgroup per CU), or some XCDs need a
        // second round (Net3x: 9 chunks x G = 28 put 36 workgroups on four XCDs of 32 CUs -- 0.67 instead of 0.48 ms per launch)
        const int per_xcd = n.max_groups / 8;
        int G = per_xcd >= L.nchunks ? 8 * (per_xcd / L.nchunks) : n.max_groups / L.nchunks;
        if (G < 1) G = 1;
        if (G > items) G = (int)items;
        a.G = G;
        a.slope = L.slope; a.scale = L.scale;
        const int dbg = n.opt.dbg;
        a.dbg = dbg;
        a.tail_w = tail_w; a.tplanes = tplanes; a.tail_form = tplanes ? tail_form : 0;
        a.tail_split = (tplanes && mixed && tail_split_for(key)) ? 1 : 0;
        a.tail1_w = tail1_w; a.tail1_out = tail1_out;
        if (pool_out && !x3 && L.r == 1 && L.nchunks == 1 && !res && pooled_groups_ok((long long)a.py, (long long)B * a.py, n.max_groups)) {      // conv3x3_rw's pooled epilogue (SEDN rblock.2)
            a.pool = pool_out; a.pool_slabs = pool_slabs;
            a.G = pooled_groups((long long)a.py, (long long)B * a.py, n.max_groups);      // its work items are patch ROWS (conv3x3_rw.hip, EPI 4): slab contents independent of the launch's plane count (common.h)
        }
        // 3x3 / 64-input-channel layers with shared weights run on the software-pipelined kernel (conv3x3_sp.hip); everything else
        // (1x1 convs, SEDN's per-plane `trans`, epilogues that kernel does not compile, MOE_CONV_IMPL=v1) on the generic one
        const bool fast = L.taps == 9 && L.nseg == 1 && !L.per_plane && n.opt.conv_impl == 2;
        if (tplanes && !(fast && !x3)) return false;
        bool fused_ok = true;
        // PReLU-only epilogues (first upsampler stage of Net4x, SEDN's rblock convs) run on the register-resident-weights kernel
        // (conv3x3_rw.hip: 6 % faster there), and so does the fused tail in its phase-class-sums form (tail_form = sums); option
        // sp_impl = sp keeps the PReLU epilogues on conv3x3_sp (A/B)
        const int rw_mode = n.opt.sp_impl;
        auto launch = [&](const ConvArgs& ca) {
            if (ca.tplanes && ca.tail_form == 1) {       // phase-class sums: conv3x3_rw is the only producer of that buffer layout
                if (!(fast && launch_conv3x3_rw(ca, s))) fused_ok = false;
                return;
            }
            if (fast && n.opt.up_impl == 1 && !ca.tplanes && !ca.pool && ca.r == 2 && ca.nchunks == 4 && ca.in_cs == 64 && ca.acc_mode == 0 && !ca.res && !ca.out_lo && ca.scale == 1.f &&
                !ca.dbg && L.has_bias && (long long)ca.B * ((ca.W + kTileW - 1) / kTileW) * (ca.H / 4) >= 32ll * n.max_groups) {
                // The x2 upsampler stages that store their tensor: all four phases in one workgroup (conv3x3_ps4.hip, store form) -- when a workgroup gets at least
                // 32 four-row blocks: a range recomputes two blocks at its ends, and a launch of three planes of 256 x 256 (the reference's own per-tile loop) would
                // give each of the 256 workgroups six.  conv3x3_rw<1> below produces the same bits (same MFMAs in the same order), so the choice is invisible.
                Ps4Args q{};
                q.in = ca.in; q.wpk = ca.wpk; q.bias = ca.bias; q.out = ca.out; q.slope = ca.slope; q.B = ca.B; q.H = ca.H; q.W = ca.W;
                if (launch_conv3x3_ps4(q, n.max_groups, s)) return;
            }
            if (fast && rw_mode && !ca.tplanes && launch_conv3x3_rw(ca, s)) { pool_done = ca.pool != nullptr; return; }
            if (fast && launch_conv3x3_sp(ca, s)) return;
            if (ca.tplanes) { fused_ok = false; return; }
            launch_conv_mfma(ca, L.taps, L.nseg, s);
        };
        // lite's 1x1 convs (conv_input2, the upsampler stages with or without the folded 48->1 tail): the HBM-bound kernel of conv1x1.hip,
        // in fp16 or with split operands; MOE_CONV1X1=0 keeps them on the generic kernel (A/B)
        const bool c1 = n.opt.conv1x1;
        if (c1 && L.taps == 1 && L.nseg == 1 && !L.per_plane && !res && L.scale == 1.f && !tplanes && n.opt.conv_impl == 2 && (!x3 || (in.lo && L.has_x3))) {
            Conv1x1Args q{};
            q.in_hi = in.hi; q.in_lo = x3 ? in.lo : nullptr; q.out_hi = out.hi; q.out_lo = x3 ? out.lo : nullptr;
            q.w_hi = blob<half_t>(L.w_hi); q.w_lo = x3 ? blob<half_t>(L.w_lo) : nullptr; q.bias = a.bias;
            q.tail_w = tail1_w; q.tail_out = tail1_out; q.slope = L.slope;
            q.B = B; q.H = H; q.W = W; q.r = L.r; q.nchunks = L.nchunks; q.out_cs = out_cs;
            q.nks = (L.cin <= 48 && n.opt.k48) ? 3 : 4;
            const int rec = prof_begin(key, (x3 ? 3 : 1) * 2.0 * (double)B * H * W * L.cout * L.cin);
            const bool ok = launch_conv1x1(q, n.max_groups, s);
            prof_end(rec);
            if (ok) { if (tail1_out) tail1_parts = 1; return true; }
        }
        if (!x3 && (dbg & 64) && fast && key == n.opt.trace_key) {   // timing trace of one launch -> /tmp/moe_trace.bin
            unsigned long long* tr = nullptr;
            const size_t nb = 8 * 32 * 4 * 16 * 8;
            if (hipMalloc((void**)&tr, nb) == hipSuccess) {
                (void)hipMemsetAsync(tr, 0, nb, s);
                ConvArgs t = a; t.acc32 = (float*)tr;
                launch(t);
                std::vector<unsigned long long> host(nb / 8);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(host.data(), tr, nb, hipMemcpyDeviceToHost);
                if (FILE* f = fopen("/tmp/moe_trace.bin", "wb")) { fwrite(host.data(), 1, nb, f); fclose(f); }
                (void)hipFree(tr);
                return true;
            }
        }
        if (!x3 && res && res->lo && out.lo) {
            // MIXED, single-pass layer on the trunk stream: fp16 operands, but the residual is read as hi + lo * 2^-11, added in fp32
            // and the sum stored as hi and lo again (the split-precision final epilogue with the residual's low part as its addend):
            // the stream x + s*conv2(...) is carried to ~22 bits through the six ARSBs, only the MFMA operand is its fp16 part
            ConvArgs q = a; q.acc_mode = 3; q.side16 = res->lo; q.out_lo = out.lo; q.res_lo = nullptr;
            const int rec = prof_begin(key, 2.0 * (double)B * H * W * L.cout * L.cin * L.taps);
            const bool ok = fast && launch_conv3x3_sp(q, s);
            prof_end(rec);
            return ok;
        }
        if (!x3) {
            const int rec = prof_begin(key, 2.0 * (double)B * H * W * L.cout * L.cin * L.taps);   // algorithmic (real channel counts)
            launch(a);
            prof_end(rec);
            return fused_ok;
        }
        if (L.has_x3 && !fast) {
            // 1x1 conv: all three products in one launch (K segments (w_lo, a_hi), (w_hi, a_lo), (w_hi, a_hi)); the activations are
            // read once per product from L2/HBM and nothing goes through the fp32 side buffer (2.6x less traffic than three passes)
            ConvArgs f4 = a; f4.wpk = blob<half_t>(L.w_x3); f4.acc_mode = 4; f4.in_lo = in.lo; f4.out_lo = out.lo; f4.res_lo = res ? res->lo : nullptr;
            launch_conv_mfma(f4, 1, 3, s);
            return true;
        }
        {   // 3x3 64->64 with both weight parts packed for it: all three products in ONE launch (conv64_x3.hip)
            if (n.opt.x3_fuse && fast && L.w_arsb_lo && in.lo && out.lo && (!res || res->lo) && !tplanes && !L.has_bias) {
                ConvX3Args q{};
                q.in_hi = in.hi; q.in_lo = in.lo; q.out_hi = out.hi; q.out_lo = out.lo;
                q.res_hi = res ? res->hi : nullptr; q.res_lo = res ? res->lo : nullptr;
                q.w_hi = blob<half_t>(L.w_arsb); q.w_lo = blob<half_t>(L.w_arsb_lo); q.zero = small<half_t>("zero");
                q.slope = L.slope; q.B = B; q.H = H; q.W = W;
                if (pool_out && !res && (L.slope == 1.f || pool_act) && pooled_groups_ok((long long)((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH),
                                                                            (long long)B * ((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH), n.max_groups)) {
                    q.pool = pool_out; q.pool_slabs = pool_slabs;      // (conv64_x3's patches are 8 x 32 outputs, as the launcher counts them)
                }
                if (gate_in && res && !use_q8()) q.gate = gate_in;
                const int rec = prof_begin(key, 3 * 2.0 * (double)B * H * W * L.cout * L.cin * L.taps);
                bool ok = false;
                // The two correction products on fp8 operands (conv64_q8.hip): 'mixed' only -- 'fp16x3' promises 2e-5, fp8 corrections deliver ~15 bits.
                // auto: the SR nets, whose all-tile sweep keeps its margin with it (worst 8.1e-4 either way, profiles/r03/m_conv64_q8.txt); the DN nets
                // (dn_lite5 7.2e-4 -> 8.6e-4 of the 1e-3 bar) stay on three fp16 products.
                if (use_q8() && L.wq_hi8 && !q.pool && (!res || res->lo8 == in.lo8)) {
                    ConvX3Args q8 = q;
                    q8.wq_hi16 = blob<half_t>(L.w_hi); q8.wq_hi8 = blob<unsigned char>(L.wq_hi8); q8.wq_lo8 = blob<unsigned char>(L.wq_lo8);
                    q8.in8 = in.lo8; q8.out8 = out.lo8;
                    if (n.opt.q8_impl == 1) ok = launch_conv64_sq(q8, n.max_groups, s);
                    if (!ok) ok = launch_conv64_q8(q8, n.max_groups, s);
                }
                if (!ok && any8) { prof_end(rec); return false; }
                if (!ok) ok = launch_conv64_x3(q, n.max_groups, s);
                prof_end(rec);
                if (ok) { pool_done = q.pool != nullptr; gate_done = q.gate != nullptr; return true; }
            }
        }
        if (fast && L.nchunks <= 16) {
            // 3x3 conv: the two low-order products run on the fast kernel as ordinary fp16-output convolutions --
            //   side = conv(w_lo, a_hi)            (plain epilogue)
            //   side = conv(w_hi, a_lo) + side     (residual epilogue, in place)
            // both in units of 2^-11; the main pass adds side * 2^-11 before its epilogue.  fp16 is plenty for a term that small,
            // and nothing goes through the fp32 side buffer (its read-modify-write traffic bounded the three-pass form).
            // (a residual's low part is in the same units: it rides along as the `residual` of the first low-order launch)
            ConvArgs q1 = a; q1.wpk = blob<half_t>(L.w_lo); q1.out = side16; q1.res = (res && res->lo) ? res->lo : nullptr; q1.bias = small<float>("zero_bias");
            q1.bias_img = small<float>("zero_bias_img"); q1.slope = 1.f; q1.scale = 1.f; q1.acc_mode = 0; q1.tail_w = nullptr; q1.tplanes = nullptr;
            ConvArgs q2 = q1; q2.in = in.lo; q2.wpk = blob<half_t>(L.w_hi); q2.res = side16;
            if (launch_conv3x3_sp(q1, s) && launch_conv3x3_sp(q2, s)) {
                ConvArgs q3 = a; q3.acc_mode = 3; q3.side16 = side16; q3.out_lo = out.lo; q3.res_lo = nullptr;   // res_lo is inside side16
                if (launch_conv3x3_sp(q3, s)) return true;
                // (a final epilogue the fast kernel does not compile, e.g. a PReLU slope above 1: redo the layer through the fp32 side buffer)
            }
        }
        // hi/lo split: (w_lo * a_hi) -> acc32,  += (w_hi * a_lo),  then (w_hi * a_hi) + acc32/2048 and the epilogue
        if (!acc32) return false;                               // (MIXED carries no fp32 side buffer)
        a.acc32 = acc32;
        ConvArgs p1 = a; p1.wpk = L.per_plane ? plane_w_lo : blob<half_t>(L.w_lo); p1.acc_mode = 1; p1.res = nullptr; p1.bias = nullptr;
        launch(p1);
        ConvArgs p2 = a; p2.in = in.lo; p2.acc_mode = 2; p2.res = nullptr; p2.bias = nullptr;
        launch(p2);
        ConvArgs p3 = a; p3.acc_mode = 3; p3.out_lo = out.lo; p3.res_lo = res ? res->lo : nullptr;
        launch(p3);
        return true;
    }
};

size_t acc32_need(const moe_net& n, int B, int h, int w)
{
    // largest [B][H][W][nchunks*64] fp32 any conv of this net produces (pre-shuffle coordinates)
    size_t best = 0;
    long long HW = (long long)h * w;
    if (n.arch == MOE_ARCH_SEDN) return (size_t)B * (size_t)HW * 256;
    int rr = 1;
    best = (size_t)B * (size_t)HW * 64;
    for (int s = 0; s < n.stages; ++s) {
        best = std::max<size_t>(best, (size_t)B * (size_t)HW * rr * rr * 64 * n.r * n.r);
        rr *= n.r;
    }
    return best;
}

int default_exact_blocks(int arch);

int exact_blocks_of(const moe_net& n)
{
    // leading ARSBs with split operands under MOE_PREC_MIXED.  Emulated error budget (tools/emu_precision.py), worst
    // of uniform-noise tiles: Net4x 7.6e-4 / 6.2e-4 / 5.1e-4 with 0 / 1 / 3 blocks; Net2x (whose trunk is 61 % of the net and whose
    // output swing is three times larger) 1.3e-3 / 9.1e-4 / 6.2e-4 / 3.7e-4 with 0 / 1 / 3 / 6; NetDN 7.8e-4 / 6.5e-4 / 5.0e-4 with 0 / 1 / 3
    if (n.exact_blocks >= 0) return n.exact_blocks > 6 ? 6 : n.exact_blocks;
    const int env = n.opt.exact_blocks_env;
    if (env >= 0) return env > 6 ? 6 : env;
    if (n.calib_valid && n.calib_blocks >= 0) return n.calib_blocks;      // measured on this checkpoint (moe_net_calibrate), never below the architecture's default
    return default_exact_blocks(n.arch);
}

int default_exact_blocks(int arch)
{
    switch (arch) {
        case MOE_ARCH_NET2X: return 4;      // measured on the GPU, worst tile of three 1080p uint8-noise frames vs the exact mode: 1.6e-3 / 1.3e-3 / 8.8e-4 / 7.0e-4 / 5.3e-4
                                            // with 1 / 2 / 3 / 4 / 6 blocks at 14.4 / 16.0 / 17.6 / 19.0 / 22.2 ms per frame (profiles/r03): 4 keeps 30 % of margin
        case MOE_ARCH_NET3X: return 2;
        case MOE_ARCH_NET4X: return 1;
        case MOE_ARCH_NETDN: return 1;
        default: return 0;
    }
}

// The fast 3x3 kernel addresses its stores, residual loads and tap planes with 32-bit BYTE offsets.  Bytes one plane of h x w
// pixels occupies in the largest tensor such a launch touches (so that planes-per-launch = 2^32 / this):
//   trunk / LR layers            128 B per pixel (64 fp16 channels; SEDN's 256-channel rblock.4 output 512 B when unfused)
//   upsampler stage k output     128 B * r^(2(k+1)) per input pixel; the LAST stage stores no tensor when the tail is fused,
//                                its nine fp32 tap planes take 36 B per output pixel instead
long long sp_bytes_per_pixel(const moe_net& n)
{
    long long per = 128;
    if (n.arch == MOE_ARCH_SEDN) per = 512;
    if (n.arch == MOE_ARCH_LITE) {
        // lite: stage k stores 128 B * 4^(k+1) per input pixel, the LAST stage (fused with the 48 -> 1 tail) stores two fp32 partial planes per branch instead -- so the largest
        // tensor a 32-bit-offset kernel (conv1x1.hip) touches is the last stage's INPUT: 128 B * 4^(stages-1).  (Round 5's kernel census found lite8's x4 -> x8 stage on the
        // generic 64-bit kernel at 27 ms a launch: 96 planes of 1024 x 1024 x 128 B overflow the offsets and nothing split the launch set.)
        for (int st = 0; st + 1 < n.stages; ++st) per *= 4;
    }
    if (n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X) {
        long long rr = 1;
        for (int st = 0; st < n.stages; ++st) {
            rr *= (long long)n.r * n.r;
            per = std::max(per, st == n.stages - 1 ? 36 * rr : 128 * rr);     // (the unfused fallback of the last stage runs on the 64-bit kernels)
        }
    }
    return per;
}
constexpr long long kSpRange = (1ll << 32) - (1ll << 16);

bool can_fuse_tail(const moe_net& n, const Fwd& f, int B, int h, int w)
{
    if (!n.opt.fuse_tail || f.x3 || f.direct || n.debug || n.opt.conv_impl != 2 || n.stages < 1) return false;
    if (!(n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X)) return false;
    long long sc = 1;
    for (int s = 0; s < n.stages; ++s) sc *= n.r;
    if ((w * (sc / n.r)) % 4 != 0) return false;            // the fused kernel stores tap planes four input columns at a time
    if (sp_bytes_per_pixel(n) * B * h * w > kSpRange) return false;    // (forward_dev keeps every launch set inside this range)
    for (const char* br : {"u", "convt_R1"}) {
        const auto it = n.conv_index.find(std::string(br) + ".up" + std::to_string(n.stages - 1));
        if (it == n.conv_index.end() || n.convs[it->second].slope > 1.f) return false;
    }
    return true;
}

int run_forward(moe_net& n, Fwd& f, const void* x, int x_dtype, long long sB, long long sH, long long sW, const long long* x_off_dev,
                void* y, int y_dtype, const long long* y_off_dev)
{
    const int B = f.B, h = f.h, w = f.w;
    const long long P = (long long)B * h * w;
    hipStream_t s = f.s;
    if (f.x3) {
        f.acc32_elems = acc32_need(n, B, h, w);
        f.acc32 = (float*)f.ar.take(f.acc32_elems * 4);
        f.side16 = (half_t*)f.ar.take(f.acc32_elems * 2 + 4096);
    } else if (f.mixed) {
        f.side16 = (half_t*)f.ar.take((size_t)P * 64 * 2 + 4096);     // split-operand layers exist at the input resolution only
        bool steep = false;                                            // a PReLU slope above 1 takes the layer off the fast kernel's final pass
        for (int i = 1; i <= exact_blocks_of(n); ++i) steep = steep || n.convs[n.conv_index.at("c1_" + std::to_string(i))].slope > 1.f;
        if (steep) { f.acc32_elems = (size_t)P * 64; f.acc32 = (float*)f.ar.take(f.acc32_elems * 4); }
    }

    auto stem = [&](const Act& out, const Act* out2 = nullptr) {
        if (f.dry()) return;
        StemArgs a{};
        if (out2) { a.w2 = f.small<float>("stem.p2"); a.out2 = out2->hi; a.out2_lo = out2->lo; }
        a.x = x; a.x_dtype = x_dtype; a.x_off = x_off_dev; a.sB = sB; a.sH = sH; a.sW = sW;
        a.w = f.small<float>("stem"); a.slope = n.scalars.at("stem_slope");
        a.out = out.hi; a.out_lo = out.lo; a.out_lo8 = out.lo8; a.B = B; a.H = h; a.W = w; a.taps = (int)n.scalars.at("stem_taps");
        launch_stem(a, s);
    };
    // forwards that run ahead of the caller's stream (moe_net_forward_ex): everything up to here touched the net's own workspace only; the kernel that writes the caller's y
    // must not overtake what the caller enqueued before this call (y's memory may have been in use by it)
    auto gate = [&]() { if (n.out_gate && !f.dry()) (void)hipStreamWaitEvent(s, n.out_gate, 0); };
    auto tail = [&](const Act* r, const Act* u, int H, int W, bool skip) {
        gate();
        if (f.dry()) return;
        TailArgs a{};
        a.in0 = r->hi; a.w0 = f.small<half_t>("tail_r");
        if (u) { a.in1 = u->hi; a.w1 = f.small<half_t>("tail_u"); }
        if (r->lo && (!u || u->lo)) {     // split operands (FP16X3; MIXED on NetDN, whose tail convs read the hi + lo stream directly)
            a.in0_lo = r->lo; a.w0_lo = f.small<half_t>("tail_r.lo");
            if (u) { a.in1_lo = u->lo; a.w1_lo = f.small<half_t>("tail_u.lo"); a.in1_lo8 = u->lo8; }
        }
        if (skip) { a.skip = x; a.skip_dtype = x_dtype; a.skip_off = x_off_dev; a.skip_sB = sB; a.skip_sH = sH; a.skip_sW = sW; }
        a.y = y; a.y_dtype = y_dtype; a.y_off = y_off_dev; a.B = B; a.H = H; a.W = W; a.taps = (int)n.scalars.at("tail_taps");
        launch_tail(a, s);
    };

    if (n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X || n.arch == MOE_ARCH_NETDN) {
        // MIXED (the default of these nets): the stem output and the trunk stream are kept as hi + lo pairs; conv_input2 and the first
        // `nx` ARSBs run with split operands (three products, ~fp32), the others with fp16 operands and the fp32-equivalent residual
        // add (see Fwd::conv).  DESIGN.md section 5 has the error budget behind this choice.
        const bool mixed = f.mixed;
        const int nx = mixed ? exact_blocks_of(n) : 0;
        Act A = f.act(P, 64, mixed), Bb = f.act(P, 64, mixed), Cc = f.act(P, 64, mixed);
        if (mixed && n.opt.conv_impl != 2)
            return fail(MOE_EINVAL, "precision 'mixed' runs on the fast 3x3 kernels only: conv_impl=v1 (MOE_CONV_IMPL=v1) is a debugging switch for 'fp16' / 'fp16x3'");
        auto trunk_conv = [&](const std::string& key, const Act& in, const Act& out, const Act* res, bool exact) {
            if (!f.conv(key, in, out, res, h, w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, exact))
                return fail(MOE_EINVAL, "layer %s: %d planes of %dx%d exceed the conv kernel's addressing range (use smaller tiles)", key.c_str(), B, h, w);
            return (int)MOE_OK;
        };
        // The chain of split-operand layers (conv_input2, the first nx ARSBs) on conv64_q8 passes its low parts as the fp8 words its correction products read
        // (Act::lo8): three quarters of the bytes of layers that are held by bytes (profiles/r03/o_stream_bytes.txt), no conversion of a_lo inside the kernel.
        // The last conv_2 writes an fp16 low part again -- into the stem's low-part buffer, which conv_input2 was the only reader of (its own buffer still
        // holds the fp8 residual it reads) -- for the fused ARSB kernels behind it.  The stem writes its low part in that form too (conv_input2 is its only
        // reader: the U branch takes the fp16 part).  Debug taps read fp16 low parts: no chain under set_debug.
        f.Hq = h; f.Wq = w;
        // NetDN (round 6): its tail convs read the stem's hi + lo pair at the very end -- tail3 takes the stem's low part as the same fp8 words (TailArgs::in1_lo8), so the
        // stem writes ONE low-part form here too.  What NetDN cannot share is the buffer: the two-launch fallback of the last exact block writes its fp16 low part into a
        // buffer of its own (lo16x) instead of over the stem's words.
        const bool dn = n.arch == MOE_ARCH_NETDN;
        bool chain8 = mixed && f.use_q8() && n.opt.lo8 && !n.debug && !f.direct && nx >= 1 && (f.dry() || (A.lo && Bb.lo && Cc.lo));      // (the planning pass has no pointers: lo16x below must be planned too)
        if (dn && (int)n.scalars.at("tail_taps") != 9) chain8 = false;      // (the 1x1 tail kernel reads fp16 pairs only)
        for (int i = 0; chain8 && i <= nx; ++i)
            for (int j = (i == 0 ? 2 : 1); chain8 && j <= 2; ++j)
                chain8 = f.q8_capable(n.convs[n.conv_index.at(i == 0 ? std::string("input2") : "c" + std::to_string(j) + "_" + std::to_string(i))]);
        half_t* lo16x = (dn && chain8) ? (half_t*)f.ar.take((size_t)P * 64 * 2 + 2048) : nullptr;
        A.lo8 = Bb.lo8 = chain8;
        stem(A);
        f.tap("stem", A, h, w, 64, n.C);
        // ---- the two upsampler branches: R on the trunk output, U on the stem output (models.py:117-123) -- planned here, because a small launch set starts its U branch NOW, on a second stream ----
        Act fin[2];
        float* tp[2] = {nullptr, nullptr};
        const bool sr = n.arch != MOE_ARCH_NETDN;
        const bool fuse = sr && can_fuse_tail(n, f, B, h, w);      // last upsampler conv + 64->1 tail conv in one kernel
        bool ps4 = false, ps9 = false;
        int H = h, W = w;
        if (sr) {   // form of the fused tail's output: phase-class sums when the last stage is a x2 shuffle the register-weight kernel takes
            int hl = h, wl = w;
            for (int st = 0; st + 1 < n.stages; ++st) { hl *= n.r; wl *= n.r; }
            bool ok = fuse && n.opt.tail_form == 1 && n.r == 2 && wl % 4 == 0 && tailsum_fits(B, hl, wl) &&
                      2ll * B * hl * wl * 64 + 2ll * (wl + 1) * 64 < (1ll << 32) - 65536;
            bool ps4_ok = true;
            for (const char* br : {"u", "convt_R1"}) {
                const auto it = n.conv_index.find(std::string(br) + ".up" + std::to_string(n.stages - 1));
                ok = ok && it != n.conv_index.end() && n.convs[it->second].slope < 1.f;
                ps4_ok = ps4_ok && it != n.conv_index.end() && ps4_tail_applicable(B, hl, wl, n.convs[it->second].slope);      // (the launcher's own predicate: common.h)
            }
            f.tail_form = ok ? 1 : 0;
            // the same layer with all four phases in one workgroup (conv3x3_ps4.hip): one fp32 plane + column aprons per branch, added by tailadd
            ps4 = ok && n.opt.up_impl == 1 && ps4_ok && (2 * wl) % 8 == 0;
            // x3 nets: the phase-row form of the same kernel (conv3x3_ps9.hip): three fp32 planes S[dy] + column aprons per branch, added by tailadd3
            ps9 = fuse && n.opt.up_impl == 1 && n.r == 3 && n.max_groups >= 3;
            for (const char* br : {"u", "convt_R1"}) {
                const auto it = n.conv_index.find(std::string(br) + ".up" + std::to_string(n.stages - 1));
                ps9 = ps9 && it != n.conv_index.end() && ps9_tail_applicable(B, hl, wl, n.convs[it->second].slope);
            }
        }
        float* ps_plane[2] = {nullptr, nullptr};
        float* ps_apron[2] = {nullptr, nullptr};
        auto run_branch = [&](int br, Act cur) -> int {      // launches go to f.s (the caller's stream, or the side stream while the U branch is forked)
            if (f.mixed) { cur.lo = nullptr; cur.lo8 = false; }      // MIXED: the upsampler convs take the fp16 parts (FP16X3 keeps its pairs)
            H = h; W = w;
            for (int st = 0; st < n.stages; ++st) {
                const std::string key = std::string(br == 0 ? "convt_R1" : "u") + ".up" + std::to_string(st);
                if (ps4 && st == n.stages - 1) {
                    ps_plane[br] = (float*)f.ar.take(ps4_plane_bytes(B, H, W) + 4096);
                    ps_apron[br] = (float*)f.ar.take(ps4_apron_bytes(B, H, W) + 4096);
                    if (!f.dry()) {
                        const ConvLayer& L = n.convs[n.conv_index.at(key)];
                        Ps4Args q{};
                        q.in = cur.hi; q.wpk = f.blob<half_t>(L.w_hi); q.bias = L.has_bias ? f.blob<float>(L.bias) : f.small<float>("zero_bias");
                        q.tail_w = f.small<half_t>(br == 0 ? "tail_r.frag" : "tail_u.frag");
                        q.plane = ps_plane[br]; q.apron = ps_apron[br]; q.slope = L.slope; q.B = B; q.H = H; q.W = W;
                        q.split = (f.mixed && f.tail_split_for(key)) ? 1 : 0;
                        const int rec = f.prof_begin(key, 2.0 * (double)B * H * W * L.cout * L.cin * L.taps);
                        bool done = false;
                        for (int rep = f.repeats(key); rep > 0; --rep) done = launch_conv3x3_ps4(q, n.max_groups, f.s);
                        f.prof_end(rec);
                        if (!done) return fail(MOE_EINVAL, "fused tail kernel (ps4) rejected layer %s", key.c_str());
                    }
                    H *= n.r; W *= n.r;
                    continue;
                }
                if (ps9 && st == n.stages - 1) {
                    ps_plane[br] = (float*)f.ar.take(ps9_plane_bytes(B, H, W) + 4096);
                    ps_apron[br] = (float*)f.ar.take(ps9_apron_bytes(B, H, W) + 4096);
                    if (!f.dry()) {
                        const ConvLayer& L = n.convs[n.conv_index.at(key)];
                        Ps9Args q{};
                        q.in = cur.hi; q.wpk = f.blob<half_t>(L.w_hi); q.bias = L.has_bias ? f.blob<float>(L.bias) : f.small<float>("zero_bias");
                        q.tail_w = f.small<half_t>(br == 0 ? "tail_r.frag" : "tail_u.frag");
                        q.plane = ps_plane[br]; q.apron = ps_apron[br]; q.slope = L.slope; q.B = B; q.H = H; q.W = W;
                        q.split = (f.mixed && f.tail_split_for(key)) ? 1 : 0;
                        const int rec = f.prof_begin(key, 2.0 * (double)B * H * W * L.cout * L.cin * L.taps);
                        bool done = false;
                        for (int rep = f.repeats(key); rep > 0; --rep) done = launch_conv3x3_ps9(q, n.max_groups, f.s);
                        f.prof_end(rec);
                        if (!done) return fail(MOE_EINVAL, "fused tail kernel (ps9) rejected layer %s", key.c_str());
                    }
                    H *= n.r; W *= n.r;
                    continue;
                }
                if (fuse && st == n.stages - 1) {
                    tp[br] = (float*)f.ar.take(f.tail_form == 1 ? (size_t)tailsum_layout(B, H, W).total * 4 + 4096 : (size_t)9 * B * H * n.r * W * n.r * 4 + 4096);
                    const half_t* frag = f.dry() ? nullptr : f.small<half_t>(br == 0 ? "tail_r.frag" : "tail_u.frag");
                    if (!f.conv(key, cur, Act{}, nullptr, H, W, nullptr, nullptr, frag, f.dry() ? (float*)16 : tp[br]))
                        return fail(MOE_EINVAL, "fused tail kernel rejected layer %s", key.c_str());
                    H *= n.r; W *= n.r;
                    continue;
                }
                Act nxt = f.act((long long)B * H * n.r * W * n.r);
                f.conv(key, cur, nxt, nullptr, H, W);
                H *= n.r; W *= n.r;
                f.tap(std::string(br == 0 ? "r" : "u") + ".up" + std::to_string(st), nxt, H, W, 64, 64);
                cur = nxt;
            }
            fin[br] = cur;
            return (int)MOE_OK;
        };
        // Small launch sets (the reference's per-tile loop hands the net 3 planes of <= 256 x 256): a launch of a few planes leaves most CUs idle while its last workgroups
        // finish (702 ARSB patches over 256 workgroups; 24 four-row blocks per ps4 workgroup, two of them recomputed), and the U branch does not depend on the trunk -- it
        // forks here onto a second stream behind the stem and joins in front of the branch sum; its workgroups take the CUs the trunk's launches leave.  Same kernels, same
        // bits (tests); the predicate is a function of the shape alone, so the workspace plan (dry run) allocates in the same order.
        // (not under FP16X3: its three-launch fallback shares one side buffer between the layers)
        const bool forked = sr && n.opt.branch_streams && !n.debug && !f.direct && !f.x3 && n.opt.conv_impl == 2 && (long long)B * h * w <= 4ll * 65536;      // (up to four planes of 256 x 256: the per-tile call
                                                                                                                                          // with or without alpha; a ragged group of 21 planes of 88 x 256 inside a
                                                                                                                                          // batched frame is NOT small: forked, the headline frame lost 0.5 ms)
        // ... each branch on its own share of the chip: side stream g_side persistent workgroups per launch, the caller's stream the rest (restored at scope exit)
        struct GroupsGuard { moe_net& n; int all; ~GroupsGuard() { n.max_groups = all; } } groups_guard{n, n.max_groups};
        // every exit behind the fork joins the side stream back into the caller's: an early return (a refused layer, a failed HIP call) must not leave kernels in flight on
        // n.side that the caller's stream never waits for -- the caller may synchronize its own stream and reuse or free the workspace (forward_dev_chunk's regrow does), and
        // an unjoined fork makes hipStreamEndCapture fail.  Disarmed by the regular join in front of the branch sum.
        struct ForkGuard {
            moe_net& n; hipStream_t s; bool armed = false;
            ~ForkGuard()
            {
                if (!armed) return;
                if (hipEventRecord(n.ev_join, n.side) != hipSuccess || hipStreamWaitEvent(s, n.ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(n.side);
            }
        } fork_guard{n, s};
        int g_side = 0;
        if (forked && n.max_groups >= 32) {
            g_side = n.opt.branch_groups > 0 ? n.opt.branch_groups : (n.stages >= 2 ? 5 * n.max_groups / 16 : 3 * n.max_groups / 16);
            g_side = std::max(8, std::min(g_side, n.max_groups - 8) / 8 * 8);
        }
        if (forked) {
            if (g_side) n.max_groups = g_side;
            if (!f.dry()) {
                if (!n.side) {
                    HIP_TRY(hipStreamCreateWithFlags(&n.side, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&n.ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&n.ev_join, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(n.ev_fork, s));
                HIP_TRY(hipStreamWaitEvent(n.side, n.ev_fork, 0));      // behind the stem (and everything before it on the caller's stream: the previous forward's reads of the workspace)
                f.s = n.side;
                fork_guard.armed = true;
            }
            const int rc = run_branch(1, A);
            if (g_side) n.max_groups = groups_guard.all - g_side;      // the trunk and the R branch: the other CUs
            if (!f.dry()) {
                f.s = s;
                if (!rc) HIP_TRY(hipEventRecord(n.ev_join, n.side));
            }
            if (rc) return rc;
        }
        if (int rc = trunk_conv("input2", A, Bb, nullptr, mixed)) return rc;
        f.tap("input2", Bb, h, w, 64, n.C);
        // single-pass ARSBs run as ONE kernel (arsb32c.hip: conv_1's output never leaves the CU) that streams cur -> oth;
        // split-operand / debug blocks use the two-launch form, conv_1 into `oth`, conv_2 back onto `cur`
        const bool arsb_fuse = n.opt.arsb_fuse && !f.x3 && !f.direct && n.opt.conv_impl == 2;
        Act cur = Bb, oth = Cc;
        for (int i = 1; i <= 6; ++i) {
            const bool ex = mixed && i <= nx;
            const std::string k1 = "c1_" + std::to_string(i), k2 = "c2_" + std::to_string(i);
            if (arsb_fuse && !ex && (!mixed || (cur.lo && oth.lo)) && n.convs[n.conv_index.at(k1)].w_arsb) {
                bool done = f.dry();
                if (!done) {
                    const ConvLayer& L1 = n.convs[n.conv_index.at(k1)];
                    const ConvLayer& L2 = n.convs[n.conv_index.at(k2)];
                    ArsbArgs q{};
                    q.x_hi = cur.hi; q.x_lo = mixed ? cur.lo : nullptr; q.y_hi = oth.hi; q.y_lo = mixed ? oth.lo : nullptr;
                    q.w1 = f.blob<half_t>(L1.w_arsb); q.w2 = f.blob<half_t>(L2.w_arsb); q.zero = f.small<half_t>("zero");
                    q.slope = L1.slope; q.B = B; q.H = h; q.W = w;
                    const bool trace = n.opt.arsb_trace;   // MOE_ARSB_TRACE with a -DARSB_TRACE build: stamps of ARSB 3 -> /tmp/arsb_trace.bin
                    const size_t tb = 8 * 16 * 4 * 40 * 8;
                    if (trace && i == 3 && hipMalloc((void**)&q.trace, tb) == hipSuccess) (void)hipMemsetAsync(q.trace, 0, tb, s);
                    const int rec = f.prof_begin("arsb" + std::to_string(i), 2.0 * 2.0 * (double)B * h * w * L1.cout * L1.cin * 9);
                    {
                        ArsbArgs q2 = q;
                        q2.w1 = f.blob<half_t>(L1.w_hi); q2.w2 = f.blob<half_t>(L2.w_hi);      // (pack_conv order; conv_2's carry the ScaleLayer factor as well)
                        q2.cin = (L1.cin == 48 && L2.cin == 48 && n.opt.k48) ? 48 : 64;      // NetDN: channels 48..63 are zeros in activations and weights
                        q2.drop_lo = (mixed && i == 6 && n.arch != MOE_ARCH_NETDN && !n.debug) ? 1 : 0;      // the upsamplers take the fp16 part: nobody reads block 6's low part
                        for (int rep = f.repeats("arsb" + std::to_string(i)); rep > 0; --rep)
                            done = launch_arsb32c(q2, n.max_groups, s);      // (false: the shape does not fit its 32-bit offsets -- the two-launch form below)
                    }
                    f.prof_end(rec);
                    if (q.trace) {
                        std::vector<unsigned long long> host(tb / 8);
                        (void)hipStreamSynchronize(s);
                        (void)hipMemcpy(host.data(), q.trace, tb, hipMemcpyDeviceToHost);
                        if (FILE* fp = fopen("/tmp/arsb_trace.bin", "wb")) { fwrite(host.data(), 1, tb, fp); fclose(fp); }
                        (void)hipFree(q.trace);
                    }
                }
                if (done) { std::swap(cur, oth); cur.lo8 = oth.lo8 = false; f.tap("arsb" + std::to_string(i), cur, h, w, 64, n.C); continue; }      // (fp16 low parts out; the scratch side's form is its next writer's)
            }
            if (ex && chain8 && n.opt.exact_fuse && n.opt.q8_impl == 1 && cur.lo8 && cur.lo && oth.lo) {
                // an exact block of the chain in ONE launch (arsb_sq.hip): x = cur (fp16 + fp8 low words) -> oth; the last exact block writes fp16 low parts (the
                // single-pass ARSB kernels behind it read those)
                bool done = f.dry();
                if (!done) {
                    const ConvLayer& L1 = n.convs[n.conv_index.at(k1)];
                    const ConvLayer& L2 = n.convs[n.conv_index.at(k2)];
                    if (L1.wq_hi8 && L2.wq_hi8 && !L1.has_bias && !L2.has_bias && L2.slope == 1.f) {
                        ArsbSqArgs q{};
                        q.x_hi = cur.hi; q.x_lo8 = (const unsigned char*)cur.lo; q.y_hi = oth.hi; q.y_lo = oth.lo;
                        q.w16[0] = f.blob<half_t>(L1.w_hi); q.wh8[0] = f.blob<unsigned char>(L1.wq_hi8); q.wl8[0] = f.blob<unsigned char>(L1.wq_lo8);
                        q.w16[1] = f.blob<half_t>(L2.w_hi); q.wh8[1] = f.blob<unsigned char>(L2.wq_hi8); q.wl8[1] = f.blob<unsigned char>(L2.wq_lo8);
                        q.slope = L1.slope; q.B = B; q.H = h; q.W = w; q.out8 = i < nx;
                        const int rec = f.prof_begin("xpair" + std::to_string(i), 2.0 * 2.0 * 3.0 * (double)B * h * w * L1.cout * L1.cin * 9);
                        done = launch_arsb_sq(q, n.max_groups, s);
                        f.prof_end(rec);
                    }
                }
                if (done) {
                    std::swap(cur, oth);
                    cur.lo8 = i < nx;
                    oth.lo8 = false;             // (scratch now: whoever writes it next decides its form)
                    f.tap("arsb" + std::to_string(i), cur, h, w, 64, n.C);
                    continue;
                }
            }
            Act m = oth;
            if (mixed && !ex) m.lo = nullptr;                    // single-pass ARSB: conv_1's output is an fp16 operand only
            m.lo8 = ex && chain8;
            Act bin = cur;
            if (mixed && !ex) bin.lo = nullptr;
            if (int rc = trunk_conv(k1, bin, m, nullptr, ex)) return rc;
            const Act resid = cur;
            if (ex && chain8 && i == nx) { cur.lo = lo16x ? lo16x : A.lo; cur.lo8 = false; }      // (see chain8)
            if (int rc = trunk_conv(k2, m, cur, &resid, ex)) return rc;
            f.tap("arsb" + std::to_string(i), cur, h, w, 64, n.C);
        }
        Bb = cur;
        if (n.arch == MOE_ARCH_NETDN) { tail(&Bb, &A, h, w, false); return MOE_OK; }
        if (mixed) { A.lo = nullptr; Bb.lo = nullptr; A.lo8 = Bb.lo8 = false; }      // the upsampler convs take the fp16 parts
        if (!forked) { if (int rc = run_branch(0, Bb)) return rc; if (int rc = run_branch(1, A)) return rc; }
        else {
            if (int rc = run_branch(0, Bb)) return rc;
            if (!f.dry()) { HIP_TRY(hipStreamWaitEvent(s, n.ev_join, 0)); fork_guard.armed = false; }      // the branch sum below reads both branches
            n.max_groups = groups_guard.all;
        }
        if (ps4) {
            if (!f.dry()) {
                TailAddArgs t{};
                t.p0 = ps_plane[0]; t.p1 = ps_plane[1]; t.a0 = ps_apron[0]; t.a1 = ps_apron[1];
                t.y = y; t.y_dtype = y_dtype; t.y_off = y_off_dev; t.B = B; t.H = H; t.W = W; t.px = (W / 2 + kTileW - 1) / kTileW;
                t.vec_ok = f.y_vec;
                const int rec = f.prof_begin("tailadd", 0.0);
                gate();
                launch_tailadd(t, s);
                f.prof_end(rec);
            }
            return MOE_OK;
        }
        if (ps9) {
            if (!f.dry()) {
                TailAdd3Args t{};
                t.p0 = ps_plane[0]; t.p1 = ps_plane[1]; t.a0 = ps_apron[0]; t.a1 = ps_apron[1];
                t.y = y; t.y_dtype = y_dtype; t.y_off = y_off_dev; t.B = B; t.H = H; t.W = W; t.px = (W / 3 + kTileW - 1) / kTileW;
                t.vec_ok = f.y_vec;
                const int rec = f.prof_begin("tailadd", 0.0);
                gate();
                launch_tailadd3(t, s);
                f.prof_end(rec);
            }
            return MOE_OK;
        }
        if (fuse) {
            if (!f.dry()) {
                TapSumArgs t{};
                t.t0 = tp[0]; t.t1 = tp[1]; t.y = y; t.y_dtype = y_dtype; t.y_off = y_off_dev; t.B = B; t.H = H; t.W = W; t.r = n.r;
                t.form = f.tail_form;
                t.vec_ok = n.r == 2 && f.y_vec && W % 8 == 0;
                gate();
                launch_tapsum(t, s);
            }
            return MOE_OK;
        }
        tail(&fin[0], &fin[1], H, W, false);
        return MOE_OK;
    }
    if (n.arch == MOE_ARCH_SEDN) {
        Act A = f.act(P), Cc = f.act(P), Dd = f.act(P), T = f.act(P, 256);
        const int nslab = (int)std::min<long long>(64, std::max<long long>(1, ((long long)h * w) / 256));
        float* partial = (float*)f.ar.take((size_t)B * nslab * 256 * 4);
        const ConvLayer& Lt = n.convs[n.conv_index.at("b0.trans")];
        const size_t wel = (size_t)Lt.nchunks * Lt.nfrag() * 512;
        half_t* wplane = (half_t*)f.ar.take(f.direct ? (size_t)B * 64 * 256 * 4 : (size_t)B * wel * 2);
        half_t* wplane_lo = f.x3 ? (half_t*)f.ar.take((size_t)B * wel * 2) : nullptr;
        stem(A);
        f.tap("stem", A, h, w, 64, 64);
        // fused block tail (see sedn_fuse in misc_kernels.hip): single-pass precision, fast kernel, planes fit the per-XCD split
        const bool sfuse = n.opt.sedn_fuse && !f.x3 && !f.direct && !n.debug && n.opt.conv_impl == 2 && B <= n.max_groups &&
                           2ll * B * h * w * 64 < (1ll << 32) - 8192;
        float* xpart = (float*)f.ar.take((size_t)B * nslab * 5 * 64 * 4);
        // the channel totals of rblock.2's output come out of that conv's epilogue (conv3x3_rw EPI 4), sedn_xsum then only visits the border
        const int pslabs = 2 * n.max_groups;
        float* xpool = (float*)f.ar.take((size_t)B * pslabs * 64 * 4);
        const bool spool = n.opt.pool_fuse;
        float* fgate = (float*)f.ar.take((size_t)B * 256 * 4);
        half_t* weff = (half_t*)f.ar.take((size_t)B * 72 * 512 * 2);
        for (int b = 0; b < 16; ++b) {
            const std::string k = "b" + std::to_string(b);
            f.conv(k + ".rb0", A, Cc, nullptr, h, w);
            f.pool_done = false;
            if (sfuse && spool && !f.dry()) { f.pool_out = xpool; f.pool_slabs = pslabs; }      // (the conv writes the first 2 min(G, py) slabs of every plane, all of them: no memset)
            f.conv(k + ".rb2", Cc, Dd, nullptr, h, w);
            f.pool_out = nullptr;
            if (sfuse) {
                if (!f.dry()) {
                    SednFuseArgs fa{};
                    fa.x = Dd.hi; fa.partial = xpart; fa.nslab = nslab; fa.B = B; fa.H = h; fa.W = w;
                    if (f.pool_done) {
                        const int py = (h + kTileH - 1) / kTileH;
                        fa.pooled = xpool; fa.pooled_slabs = pslabs;
                        fa.pooled_count = 2 * std::min(pooled_groups((long long)py, (long long)B * py, n.max_groups), py);      // conv3x3_rw EPI 4: slab 2 (g % py) + wave half
                    }
                    fa.w256t = f.small<float>(k + ".w256t"); fa.w256 = f.small<float>(k + ".w256"); fa.wt = f.small<float>(k + ".wt");
                    fa.w_down = f.small<float>(k + ".down"); fa.w_up = f.small<float>(k + ".up");
                    fa.gate = fgate; fa.weff = weff;
                    launch_sedn_fuse(fa, s);
                    ConvArgs a{};
                    a.in = Dd.hi; a.out = A.hi; a.res = A.hi; a.wpk = weff; a.plane_w = 1;
                    a.bias = f.small<float>("zero_bias"); a.bias_img = f.small<float>("zero_bias_img");
                    a.zero = f.small<half_t>("zero"); a.trash = f.small<half_t>("trash");
                    a.B = B; a.H = h; a.W = w; a.in_cs = 64; a.out_cs = 64; a.r = 1; a.nchunks = B;
                    a.px = (w + kTileW - 1) / kTileW; a.py = (h + kTileH - 1) / kTileH;
                    a.G = (int)std::max<long long>(B, std::min<long long>(n.max_groups, (long long)B * a.px * a.py));   // total workgroups (plane b gets every B-th)
                    a.slope = 0.2f; a.scale = 1.f;
                    if (!(n.opt.s64 && launch_conv64_s(a, n.max_groups, s)) && !launch_conv3x3_sp(a, s)) return fail(MOE_EINVAL, "SEDN fused block tail: kernel rejected the layer");
                }
                continue;
            }
            f.conv(k + ".rb4", Dd, T, nullptr, h, w);
            if (!f.dry()) {
                launch_pool_partial(T.hi, T.lo, partial, B, (long long)h * w, 256, nslab, s);
                const ConvLayer& L = n.convs[n.conv_index.at(k + ".trans")];
                SednSeArgs a{};
                a.partial = partial; a.nslab = nslab; a.HW = (long long)h * w;
                a.w_down = f.small<float>(k + ".down"); a.w_up = f.small<float>(k + ".up");
                a.B = B;
                if (f.direct) {
                    // plain fp32 OIHW weights [64][256]: element i -> cin = i % 256; reuse the SE kernel with a
                    // "fragment" view of 1 element per cin is not possible, so the debug path scales on the host-side layout:
                    a.trans_pk32 = f.blob<float>(L.w_plain); a.nfrag = -(64 * 256);   // negative: plain layout marker
                    a.trans_out = wplane;
                } else {
                    a.trans_pk32 = f.blob<float>(L.w_pk32); a.nfrag = L.nfrag() * L.nchunks;
                    a.trans_out = wplane; a.trans_out_lo = wplane_lo;
                }
                launch_sedn_se(a, s);
            }
            f.conv(k + ".trans", T, A, &A, h, w, wplane, wplane_lo);
            f.tap("block" + std::to_string(b), A, h, w, 64, 64);
        }
        tail(&A, nullptr, h, w, true);
        return MOE_OK;
    }
    // MOE_ARCH_LITE
    {
        Act A = f.act(P), Bb = f.act(P), Cc = f.act(P), Dd = f.act(P);
        const int nslab = (int)std::min<long long>(64, std::max<long long>(1, ((long long)h * w) / 256));
        // the pooled sums of conv_2's output come out of conv64_x3's epilogue, one slab per workgroup (fp16x3, the default of lite); in the
        // other modes a separate pass (pool_partial) forms nslab slabs per plane
        const int pslabs = n.max_groups;
        const bool poolfuse = n.opt.pool_fuse;
        float* partial = (float*)f.ar.take((size_t)B * std::max(nslab, pslabs) * 64 * 4);
        float* gate = (float*)f.ar.take((size_t)B * 64 * 4);
        // conv_input2's output is x times a fixed vector (see "stem.p2"): the stem writes it beside its own output, the 48 -> 48 1x1 conv is not launched (round 6)
        const bool stem2 = n.opt.stem2 && !n.debug && !f.direct && (int)n.scalars.at("stem_taps") == 1;
        if (stem2) stem(A, &Bb);
        else {
            stem(A);
            f.tap("stem", A, h, w, 64, 48);
            f.conv("input2", A, Bb, nullptr, h, w);
            f.tap("input2", Bb, h, w, 64, 48);
        }
        float* gate2 = (float*)f.ar.take((size_t)2 * B * 64 * 4);
        for (int k = 1; k <= 3; ++k) {
            const std::string key = "lb" + std::to_string(k);
            // round 6: conv_2 has no bias and no activation, so the mean the gate pools is linear in conv_2's INPUT: conv_1's epilogue forms the totals of its own output m,
            // frm_pre visits m's border, applies conv_2's weights to the nine shifted-window sums and the gate's two layers; conv_2 then stores gate * conv + x itself
            // (FrmPreArgs in common.h).  frm_apply's pass over six tensors (2.7 ms of a 1080p frame, at 6 TB/s) is gone.
            if (n.opt.frm_pre && poolfuse && f.x3 && !f.use_q8() && !n.debug && !f.direct && Bb.lo && Cc.lo && Dd.lo) {
                f.pool_done = false;
                if (!f.dry()) {
                    (void)hipMemsetAsync(partial, 0, (size_t)B * pslabs * 64 * 4, s);      // (workgroups without a patch in a plane leave their slab untouched)
                    f.pool_out = partial; f.pool_slabs = pslabs; f.pool_act = true;
                }
                f.conv(key + ".c1", Bb, Cc, nullptr, h, w);
                f.pool_out = nullptr; f.pool_act = false;
                if (f.dry()) { f.conv(key + ".c2", Cc, Dd, &Bb, h, w); continue; }
                if (f.pool_done) {
                    FrmPreArgs a{};
                    a.m = Cc.hi; a.m_lo = Cc.lo; a.partial = partial; a.nslab = pslabs; a.c2t = f.small<float>(key + ".c2t");
                    a.w0 = f.small<float>(key + ".w0"); a.b0 = f.small<float>(key + ".b0"); a.w2 = f.small<float>(key + ".w2"); a.b2 = f.small<float>(key + ".b2");
                    a.gate = gate2; a.B = B; a.H = h; a.W = w;
                    launch_frm_pre(a, s);
                    f.gate_in = gate2; f.gate_done = false;
                    f.conv(key + ".c2", Cc, Dd, &Bb, h, w);
                    f.gate_in = nullptr;
                    if (!f.gate_done) return fail(MOE_EINVAL, "layer %s: the gated form of conv64_x3 refused a shape its pooled form took", key.c_str());
                    std::swap(Bb, Dd);
                    f.tap(key, Bb, h, w, 64, 48);
                    continue;
                }
                // (conv_1 ran without the pooled epilogue: the shape is outside pooled_groups_ok -- the form below, from conv_2 on)
            } else
                f.conv(key + ".c1", Bb, Cc, nullptr, h, w);
            f.pool_done = false;
            if (!f.dry() && poolfuse && f.x3) {
                (void)hipMemsetAsync(partial, 0, (size_t)B * pslabs * 64 * 4, s);      // (workgroups without a patch in a plane leave their slab untouched)
                f.pool_out = partial; f.pool_slabs = pslabs;
            }
            f.conv(key + ".c2", Cc, Dd, nullptr, h, w);
            f.pool_out = nullptr;
            if (!f.dry()) {
                if (!f.pool_done) launch_pool_partial(Dd.hi, Dd.lo, partial, B, (long long)h * w, 64, nslab, s);
                FrmArgs a{};
                a.partial = partial; a.nslab = f.pool_done ? pslabs : nslab; a.HW = (long long)h * w;
                a.w0 = f.small<float>(key + ".w0"); a.b0 = f.small<float>(key + ".b0");
                a.w2 = f.small<float>(key + ".w2"); a.b2 = f.small<float>(key + ".b2");
                a.t = Dd.hi; a.x = Bb.hi; a.out = Bb.hi; a.t_lo = Dd.lo; a.x_lo = Bb.lo; a.out_lo = Bb.lo;
                a.gate = gate; a.B = B;
                launch_frm(a, s);
            }
            f.tap(key, Bb, h, w, 64, 48);
        }
        Act fin[2];
        int H = h, W = w;
        // The last upsampler stage and the 48->1 tail conv run as one kernel (the 64-channel HR tensor, 128 B per HR pixel written
        // and read back per branch, never exists): the conv's epilogue dots its fp32 activations with the tail weights.
        const bool fuse1 = n.opt.fuse_tail && !f.direct && !n.debug && n.stages >= 1;
        float* part[2] = {nullptr, nullptr};
        // fp16 input and the table of this checkpoint's U branch at hand (moe_net::lut): the U branch is not run, the final sum looks its value up
        const bool use_lut = !f.dry() && !f.lut_capture && fuse1 && x_dtype == MOE_F16 && n.lut_state == 1 && n.lut && n.opt.lite_lut;
        for (int br = 0; br < 2; ++br) {
            if (br == 1 && use_lut) { H = h * n.scale; W = w * n.scale; break; }
            Act cur = br == 0 ? Bb : A;
            H = h; W = w;
            for (int st = 0; st < n.stages; ++st) {
                const std::string ckey = std::string(br == 0 ? "ures" : "uim") + ".up" + std::to_string(st);
                if (fuse1 && f.x3 && n.opt.up_fuse2 && n.opt.conv1x1 && n.opt.k48 && n.opt.conv_impl == 2 && n.stages >= 2 && st == n.stages - 2 &&
                    128ll * B * H * W < (1ll << 32) - 65536) {      // (f.x3: every activation has its low part; no pointer tests -- the planning pass decides alike)
                    // the last TWO stages and the tail in one launch (conv1x1_f2.hip): every layer of the upsampler is pointwise -- the tensor between the stages never exists
                    const ConvLayer& LA = n.convs[n.conv_index.at(ckey)];
                    const ConvLayer& LB = n.convs[n.conv_index.at(std::string(br == 0 ? "ures" : "uim") + ".up" + std::to_string(st + 1))];
                    if (LA.taps == 1 && LB.taps == 1 && LA.cin <= 48 && LB.cin <= 48 && LA.r == 2 && LB.r == 2 && LA.nchunks == 4 && LB.nchunks == 4 && LA.w_lo && LB.w_lo &&
                        LA.slope <= 1.f && LB.slope <= 1.f && LA.scale == 1.f && LB.scale == 1.f) {
                        part[br] = (float*)f.ar.take((size_t)2 * B * H * 4 * W * 4 * 4);
                        if (!f.dry()) {
                            Conv1x1F2Args q{};
                            q.in_hi = cur.hi; q.in_lo = cur.lo;
                            q.wa_hi = f.blob<half_t>(LA.w_hi); q.wa_lo = f.blob<half_t>(LA.w_lo); q.wb_hi = f.blob<half_t>(LB.w_hi); q.wb_lo = f.blob<half_t>(LB.w_lo);
                            q.bias_a = LA.has_bias ? f.blob<float>(LA.bias) : f.small<float>("zero_bias"); q.bias_b = LB.has_bias ? f.blob<float>(LB.bias) : f.small<float>("zero_bias");
                            q.tail_w = f.small<float>(br == 0 ? "tail_r.f32" : "tail_u.f32"); q.tail_out = part[br];
                            q.slope_a = LA.slope; q.slope_b = LB.slope; q.B = B; q.H = H; q.W = W;
                            const int rec = f.prof_begin(ckey, 3.0 * 2.0 * (double)B * H * W * (LA.cout * LA.cin + 4.0 * LB.cout * LB.cin));
                            const bool ok = launch_conv1x1_f2(q, n.max_groups, s);
                            f.prof_end(rec);
                            if (!ok) return fail(MOE_EINVAL, "fused upsampler stages (conv1x1_f2) rejected layer %s", ckey.c_str());
                            f.tail1_parts = 1;
                        }
                        H *= 4; W *= 4;
                        break;
                    }
                }
                if (fuse1 && st == n.stages - 1) {
                    part[br] = (float*)f.ar.take((size_t)2 * B * H * 2 * W * 2 * 4);
                    f.conv(ckey, cur, Act{}, nullptr, H, W, nullptr, nullptr, nullptr, nullptr,
                           f.dry() ? nullptr : f.small<float>(br == 0 ? "tail_r.f32" : "tail_u.f32"), f.dry() ? (float*)16 : part[br]);
                    H *= 2; W *= 2;
                    continue;
                }
                Act nxt = f.act((long long)B * H * 2 * W * 2);
                f.conv(ckey, cur, nxt, nullptr, H, W);
                H *= 2; W *= 2;
                f.tap(std::string(br == 0 ? "r" : "u") + ".up" + std::to_string(st), nxt, H, W, 64, 48);
                cur = nxt;
            }
            fin[br] = cur;
        }
        if (fuse1) {
            if (!f.dry()) {
                if (f.lut_capture) {      // (B = 1, the 256 x 256 image of all patterns: part[1] IS the table)
                    if (f.tail1_parts != 1 || !part[1]) return MOE_EINVAL;
                    HIP_TRY(hipMemcpyAsync(f.lut_capture, part[1], (size_t)H * W * 4, hipMemcpyDeviceToDevice, s));
                    return MOE_OK;
                }
                if (use_lut && f.tail1_parts != 1) return fail(MOE_EINVAL, "lite: the U-branch table needs the one-part form of the fused tail");
                Tail1SumArgs t{};
                t.p0 = part[0]; t.p1 = use_lut ? nullptr : part[1]; t.nparts = f.tail1_parts; t.y = y; t.y_dtype = y_dtype; t.y_off = y_off_dev; t.B = B; t.H = H; t.W = W;
                if (use_lut) { t.lut = n.lut; t.r = n.scale; t.x = x; t.x_off = x_off_dev; t.sB = sB; t.sH = sH; t.sW = sW; t.vec_ok = f.y_vec; }
                launch_tail1sum(t, s);
            }
            return MOE_OK;
        }
        if (f.lut_capture) return MOE_EINVAL;      // (no fused tail: no table -- build_lite_lut marks it unavailable)
        tail(&fin[0], &fin[1], H, W, false);
    }
    return MOE_OK;
}

size_t workspace_need(moe_net& n, int B, int h, int w)
{
    Fwd f{n, nullptr, B, h, w, Arena{}, n.precision == MOE_PREC_FP16X3, n.precision == MOE_PREC_DEBUG_DIRECT};
    f.mixed = n.precision == MOE_PREC_MIXED;
    run_forward(n, f, nullptr, MOE_F32, 0, 0, 0, nullptr, nullptr, MOE_F32, nullptr);
    return f.ar.off + 4096;
}

int forward_dev_chunk(moe_net& n, const void* x, int x_dtype, int B, int h, int w, long long sB, long long sH, long long sW,
                      const long long* x_off_dev, void* y, int y_dtype, const long long* y_off_dev, hipStream_t s, bool y_off_mult8);

// The table of lite's U branch (moe_net::lut), filled on the first fp16 forward of a checkpoint: one ordinary forward of the net on a 256 x 256 one-plane image whose pixel
// (i, j) holds the fp16 bit pattern 256 i + j; run_forward copies the U branch's plane (part[1]) instead of summing.  Not during stream capture (it allocates), not for
// precisions / options whose fused tail is not the one-part form (then the table stays unavailable and the branch is computed as before).
static void build_lite_lut(moe_net& n, hipStream_t s)
{
    n.lut_state = -1;
    if (n.arch != MOE_ARCH_LITE || !n.opt.lite_lut || n.debug || !n.opt.fuse_tail || n.precision == MOE_PREC_DEBUG_DIRECT) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); n.lut_state = 0; return; }      // (another forward may try)
    const int r = n.scale;
    std::vector<unsigned short> pat(65536);
    for (int i = 0; i < 65536; ++i) pat[(size_t)i] = (unsigned short)i;
    if (hipMalloc((void**)&n.lut_in, 65536 * 2) != hipSuccess || hipMalloc((void**)&n.lut, (size_t)65536 * r * r * 4) != hipSuccess) { (void)hipGetLastError(); drop_lut(n); n.lut_state = -1; return; }
    if (hipMemcpyAsync(n.lut_in, pat.data(), 65536 * 2, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); drop_lut(n); n.lut_state = -1; return; }
    float* scratch = nullptr;
    if (hipMalloc((void**)&scratch, (size_t)65536 * r * r * 4) != hipSuccess) { (void)hipGetLastError(); drop_lut(n); n.lut_state = -1; return; }
    n.lut_state = 2;
    const int rc = forward_dev_chunk(n, n.lut_in, MOE_F16, 1, 256, 256, 65536, 256, 1, nullptr, scratch, MOE_F32, nullptr, s, true);
    (void)hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (rc != MOE_OK) { (void)hipGetLastError(); drop_lut(n); n.lut_state = -1; return; }
    n.lut_state = 1;
}

int forward_dev(moe_net& n, const void* x, int x_dtype, int B, int h, int w, long long sB, long long sH, long long sW,
                const long long* x_off_dev, void* y, int y_dtype, const long long* y_off_dev, hipStream_t s, bool y_off_mult8 = true)
{
    if (!n.finalized) return fail(MOE_ESTATE, "moe_net_forward: net is not finalized (load_state_dict + to(device) first)");
    if (n.arch == MOE_ARCH_LITE && x_dtype == MOE_F16 && n.lut_state == 0 && n.opt.lite_lut) build_lite_lut(n, s);
    if (B < 1 || h < 1 || w < 1) return fail(MOE_EINVAL, "moe_net_forward: bad shape B=%d h=%d w=%d", B, h, w);
    if ((x_dtype != MOE_F32 && x_dtype != MOE_F16) || (y_dtype != MOE_F32 && y_dtype != MOE_F16))
        return fail(MOE_EINVAL, "moe_net_forward: x/y dtype must be MOE_F32 or MOE_F16");
    // planes per launch set: whatever the caller batched (whole-image tiles under cropsize 'auto', MOE_TILES_PER_BATCH ...), a launch
    // never leaves the fast kernel's addressing range -- larger batches are run as several launch sets
    const long long bmax = kSpRange / (sp_bytes_per_pixel(n) * (long long)h * w);
    if (bmax < 1)
        return fail(MOE_ENOMEM, "a %dx%d tile exceeds the convolution kernels' addressing range (%lld pixels per plane at most for this net): use a smaller cropsize",
                    h, w, kSpRange / sp_bytes_per_pixel(n));
    if (B <= bmax) return forward_dev_chunk(n, x, x_dtype, B, h, w, sB, sH, sW, x_off_dev, y, y_dtype, y_off_dev, s, y_off_mult8);
    const size_t xe = x_dtype == MOE_F32 ? 4 : 2, ye = y_dtype == MOE_F32 ? 4 : 2;
    const long long yplane = (long long)h * n.scale * w * n.scale;
    for (int b0 = 0; b0 < B; b0 += (int)bmax) {
        const int cnt = (int)std::min<long long>(bmax, B - b0);
        const void* xc = x_off_dev ? x : (const void*)((const char*)x + (size_t)b0 * sB * xe);
        void* yc = y_off_dev ? y : (void*)((char*)y + (size_t)b0 * yplane * ye);
        const bool al = y_off_mult8 && (y_off_dev || ((size_t)b0 * yplane * ye) % 16 == 0);
        int rc = forward_dev_chunk(n, xc, x_dtype, cnt, h, w, sB, sH, sW, x_off_dev ? x_off_dev + b0 : nullptr, yc, y_dtype,
                                   y_off_dev ? y_off_dev + b0 : nullptr, s, al);
        if (rc) return rc;
    }
    return MOE_OK;
}

int forward_dev_chunk(moe_net& n, const void* x, int x_dtype, int B, int h, int w, long long sB, long long sH, long long sW,
                      const long long* x_off_dev, void* y, int y_dtype, const long long* y_off_dev, hipStream_t s, bool y_off_mult8)
{
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != n.device) HIP_TRY(hipSetDevice(n.device));
    const size_t need = workspace_need(n, B, h, w);
    if (need > n.ws_bytes) {
        if (n.ws) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(n.ws)); n.ws = nullptr; n.ws_bytes = 0; }
        hipError_t e = hipMalloc((void**)&n.ws, need);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MOE_ENOMEM, "workspace of %zu bytes for %d planes of %dx%d does not fit", need, B, h, w); }
        n.ws_bytes = need;
    }
    Fwd f{n, s, B, h, w, Arena{n.ws, 0}, n.precision == MOE_PREC_FP16X3, n.precision == MOE_PREC_DEBUG_DIRECT};
    f.mixed = n.precision == MOE_PREC_MIXED;
    f.y_vec = y_off_mult8 && ((uintptr_t)y % 16 == 0);   // every output plane starts 16-byte aligned: wide stores allowed
    if (n.lut_state == 2 && x == (const void*)n.lut_in) f.lut_capture = n.lut;      // (build_lite_lut's own forward)
    int rc = run_forward(n, f, x, x_dtype, sB, sH, sW, x_off_dev, y, y_dtype, y_off_dev);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

// ---- calibration of one checkpoint (moe_net_calibrate) ---------------------------------------------------------------------------------
// The per-architecture counts of split-operand ARSBs were chosen on the zoo's weights with ~2e-4 of the 1e-3 budget to spare on uint8 noise; a checkpoint
// whose trunk swings wider spends more (profiles/r04/n_margin_sweep_and_fuzz_final_tree.txt: the trunk's weights x 1.15 -> 1.3e-3 / 1.8e-3 with the defaults).
// The reference's contract is "load any state dict, get the fp32 answer" (python/imageProcess.py:319-334), so the count is a property of the CHECKPOINT and is
// measured when it is loaded: uniform uint8-noise tiles (the input class that spends the most: SURVEY.md appendix) go through the exact mode (FP16X3: pinned to the
// fp32 oracle at 2e-5 by the tests) and through MIXED with n = default .. 6 blocks; the smallest n whose PREDICTED worst tile of a full frame is within `target`
// is kept, and when not even six blocks reach it the net runs in FP16X3.  ~0.2-0.5 s once per checkpoint and device.
//
// Round 6: the measurement is conservative by construction.  Rounds 4-5 measured two tiles of 3 x 192 x 192 from one seed and compared the value
// itself with 7.5e-4; the worst tile of an all-tile sweep over full frames lay 1.05-1.29x above such a sample (a maximum over 5-20x more values, and tiles differ: a4 as
// shipped 6.7e-4 there, 7.4e-4 over 48 plane-tiles; a perturbed checkpoint 6.5e-4 on nine planes, 8.3e-4 over 48 -- profiles/r05/margin_sweep.txt,
// profiles/r06/calibration_vs_frames.txt).  Now: TWELVE seeds, tiles of 3 x 256 x 256 (the tile size that ships; 36 plane-tiles, run as four forwards of nine planes and
// compared on the device), and the comparison is  measured x kCalibInflate <= target  for what is left between this sample and the 120 plane-tiles of a 1080p frame.
// `err` reports the predicted (inflated) figure.  Hysteresis: the ARCHITECTURE'S DEFAULT count is kept while its prediction is within 5 % above the target -- a zoo key
// that sits next to the target must not flip between n and n + 1 (and change its bits and speed) with the driver, the device or the launch geometry; every
// larger count must be strictly within the target.
constexpr double kCalibTarget = 8.5e-4;       // predicted worst tile of a full frame; 1.5e-4 of the 1e-3 contract stay in hand (the exact mode itself is pinned to the oracle at 2e-5,
                                              // an fp16 result adds half an ulp of the value)
constexpr double kCalibInflate = 1.10;        // full-frame worst tile / this sample's worst value
constexpr double kCalibInflateDN = 1.30;      // ... NetDN: its all-tile sweeps lie 1.06-1.275x above the sample (dn_lite5 as shipped: 6.40e-4 on the sample, 8.16e-4 over 48 plane-tiles; profiles/r06/margin_sweep.txt)
constexpr double kCalibHysteresis = 1.05;     // the default count only
constexpr int kCalibTiles = 12;               // noise seeds = tiles of 3 planes, run in chunks of three tiles

bool calibratable(const moe_net& n) { return n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X || n.arch == MOE_ARCH_NETDN; }

int calibrate_blocks(moe_net& n, double target, hipStream_t s)
{
    if (!(target > 0)) target = kCalibTarget;
    const int B = 9, h = 256, w = 256, sc = n.scale;              // a chunk: three tiles of three planes (what one forward takes: the workspace stays that of a small launch set)
    const int nchunks = kCalibTiles / 3;
    const size_t nin = (size_t)B * h * w, nout = nin * sc * sc, per_seed = (size_t)3 * h * w;
    std::vector<float> x(nin * nchunks);
    for (int t = 0; t < kCalibTiles; ++t) {
        unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * t + 1);      // splitmix64 -> bytes -> / 255: the uint8 noise of SURVEY 8(d), one fixed seed per tile
        for (size_t i = 0; i < per_seed; i += 8) {
            unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            for (size_t k = 0; k < 8 && i + k < per_seed; ++k) x[t * per_seed + i + k] = (float)((z >> (8 * k)) & 255) / 255.f;
        }
    }
    HIP_TRY(hipSetDevice(n.device));
    // device scratch: the inputs, the exact mode's results of every chunk, one chunk of candidate results, the folded maximum -- nothing but 4 bytes per count comes back
    struct Scratch { float *xd = nullptr, *ref = nullptr, *got = nullptr; unsigned* mx = nullptr; ~Scratch() { (void)hipFree(xd); (void)hipFree(ref); (void)hipFree(got); (void)hipFree(mx); } } sc_;
    if (hipMalloc((void**)&sc_.xd, x.size() * 4) != hipSuccess || hipMalloc((void**)&sc_.ref, nout * nchunks * 4) != hipSuccess || hipMalloc((void**)&sc_.got, nout * 4) != hipSuccess ||
        hipMalloc((void**)&sc_.mx, 4) != hipSuccess) {
        (void)hipGetLastError();
        return fail(MOE_ENOMEM, "moe_net_calibrate: %zu bytes of scratch do not fit", (x.size() + nout * (nchunks + 1)) * 4);
    }
    HIP_TRY(hipMemcpyAsync(sc_.xd, x.data(), x.size() * 4, hipMemcpyHostToDevice, s));
    const int prec0 = n.precision, blocks0 = n.exact_blocks;
    const bool debug0 = n.debug;
    n.debug = false;
    auto run = [&](int c, float* out) -> int { return forward_dev(n, sc_.xd + (size_t)c * nin, MOE_F32, B, h, w, (long long)h * w, w, 1, nullptr, out, MOE_F32, nullptr, s); };
    auto restore = [&](int rc) {
        n.exact_blocks = blocks0; n.debug = debug0;
        // the measurement's launch sets are larger than a per-tile caller's (nine planes): give the workspace back, the next forward sizes it for what the caller runs
        (void)hipStreamSynchronize(s);
        if (n.ws) { (void)hipFree(n.ws); n.ws = nullptr; n.ws_bytes = 0; }
        return rc;
    };
    int rc = build_device_weights(n, MOE_PREC_FP16X3);
    if (rc) return restore(rc);
    n.precision = MOE_PREC_FP16X3;
    for (int c = 0; c < nchunks && !rc; ++c) rc = run(c, sc_.ref + (size_t)c * nout);
    if (rc) { n.precision = prec0; (void)build_device_weights(n, prec0); return restore(rc); }
    if ((rc = build_device_weights(n, MOE_PREC_MIXED))) return restore(rc);
    n.precision = MOE_PREC_MIXED;
    n.calib_valid = false; n.calib_blocks = -1; n.calib_err = 0.0;
    int best = -1;
    double err = 0.0;
    const int nb0 = default_exact_blocks(n.arch);
    for (int nb = nb0; nb <= 6; ++nb) {
        n.exact_blocks = nb;
        HIP_TRY(hipMemsetAsync(sc_.mx, 0, 4, s));
        for (int c = 0; c < nchunks && !rc; ++c) {
            rc = run(c, sc_.got);
            if (!rc) launch_maxabsdiff(sc_.got, sc_.ref + (size_t)c * nout, (long long)nout, sc_.mx, s);
        }
        if (rc) break;
        unsigned bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, sc_.mx, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        float e;
        memcpy(&e, &bits, 4);
        err = (double)e * (n.arch == MOE_ARCH_NETDN ? kCalibInflateDN : kCalibInflate);                          // the predicted worst tile of a full frame (a NaN / Inf result: +inf, no count passes)
        if (n.opt.calib_log) fprintf(stderr, "moe_net_calibrate: %d split blocks: measured %.3e on %d noise tiles of 3 x %d x %d, predicted %.3e (target %.3e)\n", nb, (double)e, kCalibTiles, h, w, err, target);
        if (err <= (nb == nb0 ? target * kCalibHysteresis : target)) { best = nb; break; }
    }
    if (!rc) { n.calib_valid = true; n.calib_blocks = best; n.calib_err = err; }
    if (prec0 != MOE_PREC_MIXED) { n.precision = prec0; const int rc2 = build_device_weights(n, prec0); if (!rc) rc = rc2; }
    return restore(rc);
}

}  // namespace

// =====================================================================================================
// plan device cache + stitch + run
// =====================================================================================================
static int plan_device_tables(const Plan& p, int device, int C, int64_t sC, int64_t sH, int64_t sW, int si, int scnt,
                              PlanDeviceCache** outp)
{
    if (scnt < 1) { scnt = 1; si = 0; }
    for (auto& up : p.dev) {
        PlanDeviceCache& d = *up;
        if (d.blob && d.device == device && d.C == C && d.sC == sC && d.sH == sH && d.sW == sW && d.shard_index == si && d.shard_count == scnt) {
            *outp = &d;
            return MOE_OK;
        }
    }
    if (p.dev.size() >= 16) {   // bounded: drop the oldest layout
        if (p.dev.front()->blob) (void)hipFree(p.dev.front()->blob);
        for (auto& e : p.dev.front()->strip_tabs) (void)hipFree(e.second);
        p.dev.erase(p.dev.begin());
    }
    p.dev.push_back(std::make_unique<PlanDeviceCache>());
    PlanDeviceCache& d = *p.dev.back();
    const size_t nt = p.tiles.size();
    std::vector<long long> xo, yo;
    int slot = 0;
    for (const auto& g : p.groups) {
        d.group_first.push_back(slot);
        int cnt = 0;
        for (int k : g.tiles) {
            if (k % scnt != si) continue;
            const TileRect& t = p.tiles[k];
            const long long plane = (long long)(g.th * p.sc) * (g.tw * p.sc);
            for (int c = 0; c < C; ++c) {
                xo.push_back((long long)c * sC + (long long)t.top * sH + (long long)t.left * sW);
                yo.push_back(p.tile_off[k] / p.C * C + (long long)c * plane);
            }
            ++slot; ++cnt;
        }
        d.group_count.push_back(cnt);
    }
    if (xo.empty()) { xo.push_back(0); yo.push_back(0); }
    d.y_mult8 = true;
    for (long long v : yo) d.y_mult8 = d.y_mult8 && (v % 8 == 0);
    // tile_off scaled to C planes (C may differ from the planning shape's channel count, e.g. alpha stripped)
    std::vector<long long> toff(nt);
    for (size_t k = 0; k < nt; ++k) toff[k] = p.tile_off[k] / p.C * C;
    std::vector<char> host;
    auto put = [&](const void* src, size_t bytes) { const size_t a = (host.size() + 255) & ~(size_t)255; host.resize(a + bytes); memcpy(host.data() + a, src, bytes); return a; };
    const size_t o_x = put(xo.data(), xo.size() * 8), o_y = put(yo.data(), yo.size() * 8), o_t = put(toff.data(), toff.size() * 8);
    const size_t o_rf = put(p.row_first.data(), p.row_first.size() * 4), o_rc = put(p.row_cnt.data(), p.row_cnt.size() * 4);
    const size_t o_cf = put(p.col_first.data(), p.col_first.size() * 4), o_cc = put(p.col_cnt.data(), p.col_cnt.size() * 4);
    const size_t o_rt = put(p.row_tab.data(), p.row_tab.size() * 4), o_ct = put(p.col_tab.data(), p.col_tab.size() * 4);
    const float zero = 0.f;
    const size_t o_rp = put(p.ramp.empty() ? &zero : p.ramp.data(), std::max<size_t>(4, p.ramp.size() * 4));
    HIP_TRY(hipMalloc(&d.blob, host.size()));
    HIP_TRY(hipMemcpy(d.blob, host.data(), host.size(), hipMemcpyHostToDevice));
    char* b = (char*)d.blob;
    d.x_off = (long long*)(b + o_x); d.y_off = (long long*)(b + o_y); d.tile_off = (long long*)(b + o_t);
    d.row_first = (int*)(b + o_rf); d.row_cnt = (int*)(b + o_rc); d.col_first = (int*)(b + o_cf); d.col_cnt = (int*)(b + o_cc);
    d.row_tab = (int*)(b + o_rt); d.col_tab = (int*)(b + o_ct); d.ramp = (float*)(b + o_rp);
    d.device = device; d.C = C; d.sC = sC; d.sH = sH; d.sW = sW; d.shard_index = si; d.shard_count = scnt;
    *outp = &d;
    return MOE_OK;
}

static void fill_stitch(const Plan& p, const PlanDeviceCache& d, StitchArgs& a, const float* tiles, const long long* tile_off, int C, void* out, int out_dtype)
{
    a.tiles = tiles; a.tile_off = tile_off;
    a.row_first = d.row_first; a.row_cnt = d.row_cnt; a.col_first = d.col_first; a.col_cnt = d.col_cnt;
    a.row_tab = d.row_tab; a.col_tab = d.col_tab; a.ramp = d.ramp;
    a.out = out; a.out_dtype = out_dtype; a.C = C; a.out_h = p.out_h; a.out_w = p.out_w; a.step_w = p.aw.step;
    a.y0 = 0; a.rows = p.out_h; a.row_lo = 0;
}

// =====================================================================================================
// extern "C"
// =====================================================================================================
struct moe_plan { Plan p; };

extern "C" {

const char* moe_last_error(void) { return g_err.c_str(); }
int moe_abi_version(void) { return MOE_ABI_VERSION; }
int moe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int moe_net_create(int arch, int scale, moe_net** out)
{
    if (!out) return fail(MOE_EINVAL, "moe_net_create: out is NULL");
    auto n = std::make_unique<moe_net>();
    n->arch = arch;
    n->opt.from_env();
    switch (arch) {
        case MOE_ARCH_NET2X: n->scale = 2; n->stages = 1; n->r = 2; break;
        case MOE_ARCH_NET3X: n->scale = 3; n->stages = 1; n->r = 3; break;
        case MOE_ARCH_NET4X: n->scale = 4; n->stages = 2; n->r = 2; break;
        case MOE_ARCH_NETDN: n->scale = 1; n->stages = 0; n->C = 48; break;
        case MOE_ARCH_SEDN: n->scale = 1; n->stages = 0; break;
        case MOE_ARCH_LITE:
            if (scale != 2 && scale != 4 && scale != 8) return fail(MOE_EINVAL, "MoeNet_lite2.Net: upscale must be 2, 4 or 8 (got %d)", scale);
            n->scale = scale; n->r = 2; n->C = 48;
            n->stages = scale == 2 ? 1 : (scale == 4 ? 2 : 3);
            break;
        default: return fail(MOE_EINVAL, "moe_net_create: unknown architecture %d", arch);
    }
    if (arch != MOE_ARCH_LITE && scale != 0 && scale != n->scale)
        return fail(MOE_EINVAL, "moe_net_create: architecture %d has scale %d, not %d", arch, n->scale, scale);
    declare_params(*n);
    *out = n.release();
    return MOE_OK;
}

static void pipe_destroy(moe_net& n);

void moe_net_destroy(moe_net* n)
{
    if (!n) return;
    drop_lut(*n);
    if (n->blob) (void)hipFree(n->blob);
    if (n->ws) (void)hipFree(n->ws);
    for (auto& t : n->taps) if (t.second.dev) (void)hipFree(t.second.dev);
    for (auto& ev : n->prof_ev) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }
    pipe_destroy(*n);
    if (n->side) (void)hipStreamDestroy(n->side);
    if (n->ev_fork) (void)hipEventDestroy(n->ev_fork);
    if (n->ev_join) (void)hipEventDestroy(n->ev_join);
    for (auto& sl : n->off_ring) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    delete n;
}

int moe_net_scale(const moe_net* n) { return n ? n->scale : 0; }
int moe_net_num_params(const moe_net* n) { return n ? (int)n->params.size() : 0; }

int moe_net_param_info(const moe_net* n, int i, const char** name, int64_t shape[4], int* ndim)
{
    if (!n || i < 0 || i >= (int)n->params.size()) return fail(MOE_EINVAL, "moe_net_param_info: index out of range");
    const Param& p = n->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape) for (size_t d = 0; d < p.shape.size() && d < 4; ++d) shape[d] = p.shape[d];
    return MOE_OK;
}

int moe_net_set_param(moe_net* n, const char* name, const float* data, const int64_t* shape, int ndim)
{
    if (!n || !name || !data) return fail(MOE_EINVAL, "moe_net_set_param: NULL argument");
    auto it = n->index.find(name);
    if (it == n->index.end()) return fail(MOE_EINVAL, "Unexpected key(s) in state_dict: \"%s\"", name);
    Param& p = n->params[it->second];
    bool same = (int)p.shape.size() == ndim;
    for (int d = 0; same && d < ndim; ++d) same = p.shape[d] == shape[d];
    if (!same) {
        std::string got, want;
        for (int d = 0; d < ndim; ++d) got += (d ? ", " : "") + std::to_string((long long)shape[d]);
        for (size_t d = 0; d < p.shape.size(); ++d) want += (d ? ", " : "") + std::to_string((long long)p.shape[d]);
        return fail(MOE_EINVAL, "size mismatch for %s: copying a param with shape (%s), the shape in current model is (%s)", name, got.c_str(), want.c_str());
    }
    p.data.assign(data, data + p.numel());
    p.set = true;
    n->finalized = false;
    n->calib_valid = false;          // (a measurement of other weights)
    n->auto_resolved = -1;
    return MOE_OK;
}

// MOE_PREC_AUTO: the arithmetic that holds the product's tolerance (1e-3 max-abs against the reference's fp32 CPU path, on natural images and on white
// noise) for the family -- the per-family policy lives HERE, behind the C ABI, so that the one-line stub of INTEGRATION.md works for every plugin-table key
static int resolve_precision(int arch, int precision)
{
    if (precision != MOE_PREC_AUTO) return precision;
    switch (arch) {
        case MOE_ARCH_SEDN: return MOE_PREC_FP16;       // l15 / l25 / l50: 4-6e-4 in plain fp16
        case MOE_ARCH_LITE: return MOE_PREC_FP16X3;     // lite2 / 4 / 8: every conv but one would have to be split anyway
        default: return MOE_PREC_MIXED;                 // Net2x / 3x / 4x, NetDN
    }
}

int moe_net_resolved_precision(const moe_net* n, int precision)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_resolved_precision: NULL net");
    if (precision < MOE_PREC_FP16 || precision > MOE_PREC_AUTO) return fail(MOE_EINVAL, "moe_net_resolved_precision: unknown precision %d", precision);
    if (precision == MOE_PREC_AUTO && n->auto_resolved >= 0) return n->auto_resolved;      // (what the last finalize with AUTO settled on for these weights)
    return resolve_precision(n->arch, precision);
}

int moe_net_finalize(moe_net* n, int device, int precision)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_finalize: NULL net");
    const bool autop = precision == MOE_PREC_AUTO;
    if (precision == MOE_PREC_AUTO) precision = resolve_precision(n->arch, precision);
    if (precision != MOE_PREC_FP16 && precision != MOE_PREC_FP16X3 && precision != MOE_PREC_DEBUG_DIRECT && precision != MOE_PREC_MIXED)
        return fail(MOE_EINVAL, "moe_net_finalize: unknown precision %d", precision);
    if (precision == MOE_PREC_MIXED && (n->arch == MOE_ARCH_SEDN || n->arch == MOE_ARCH_LITE))
        return fail(MOE_EINVAL, "moe_net_finalize: MOE_PREC_MIXED is defined for Net2x/3x/4x and NetDN (use FP16 for SEDN, FP16X3 for lite)");
    std::string missing;
    for (const auto& p : n->params) if (!p.set) missing += (missing.empty() ? "\"" : ", \"") + p.name + "\"";
    if (!missing.empty()) return fail(MOE_ESTATE, "Missing key(s) in state_dict: %s", missing.c_str());
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) { (void)hipGetLastError(); return fail(MOE_EHIP, "no HIP device available (this engine has no CPU path)"); }
    if (device < 0 || device >= cnt) return fail(MOE_EINVAL, "device %d out of range (%d visible)", device, cnt);
    if (n->device >= 0 && n->device != device) {
        // the net moves to another device: its streams, events and workspace belong to the old one (the side stream of small launch sets was created once and
        // would have been used on the new device's forwards); the weight blob is rebuilt below
        (void)hipSetDevice(n->device);
        pipe_destroy(*n);
        if (n->side) { (void)hipStreamSynchronize(n->side); (void)hipStreamDestroy(n->side); n->side = nullptr; }
        if (n->ev_fork) { (void)hipEventDestroy(n->ev_fork); n->ev_fork = nullptr; }
        if (n->ev_join) { (void)hipEventDestroy(n->ev_join); n->ev_join = nullptr; }
        if (n->ws) { (void)hipDeviceSynchronize(); (void)hipFree(n->ws); n->ws = nullptr; n->ws_bytes = 0; }
        if (n->blob) { (void)hipFree(n->blob); n->blob = nullptr; n->blob_bytes = 0; }
        for (auto& ev : n->prof_ev) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }
        n->prof_ev.clear(); n->prof_used = 0;
        for (auto& sl : n->off_ring) {
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
            if (sl.done) (void)hipEventDestroy(sl.done);
            sl = moe_net::OffSlot{};
        }
        for (auto& t : n->taps) if (t.second.dev) (void)hipFree(t.second.dev);
        n->taps.clear();
        n->calib_valid = false;      // (measured on the other device: the arithmetic is the same, but the rule is one measurement per checkpoint AND device)
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(conv_mfma_init());
    n->max_groups = conv_mfma_max_groups();
    if (n->opt.max_groups > 0) n->max_groups = n->opt.max_groups;
    n->device = device;
    n->precision = precision;
    int rc = build_device_weights(*n, precision);
    if (rc) return rc;
    n->finalized = true;
    if (autop) {
        // MOE_PREC_AUTO promises the 1e-3 contract for THIS checkpoint, not for the zoo's: the ARSB nets measure their count of split-operand blocks on the device
        // now (calibrate_blocks above; once per checkpoint -- the result is kept until a parameter changes), and fall back to FP16X3 when no count reaches the target.
        // An explicit moe_net_set_exact_blocks / MOE_EXACT_BLOCKS or option auto_calibrate = 0 leaves the per-architecture default in force.
        if (precision == MOE_PREC_MIXED && calibratable(*n) && n->opt.auto_calibrate && n->exact_blocks < 0 && n->opt.exact_blocks_env < 0) {
            if (!n->calib_valid) {
                // the measurement's device work (three weight rebuilds, up to seven forwards) runs on a stream of its own: the legacy NULL stream would serialise against
                // every blocking stream of the process and is illegal while another stream captures; callers with a stream of theirs use moe_net_calibrate
                hipStream_t cs = nullptr;
                HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                rc = calibrate_blocks(*n, 0.0, cs);
                (void)hipStreamSynchronize(cs);
                (void)hipStreamDestroy(cs);
                if (rc) { n->finalized = false; return rc; }
            }
            if (n->calib_blocks < 0) {
                rc = build_device_weights(*n, MOE_PREC_FP16X3);
                if (rc) { n->finalized = false; return rc; }
                n->precision = precision = MOE_PREC_FP16X3;
            }
        }
        n->auto_resolved = precision;
    }
    return MOE_OK;
}

int moe_net_calibrate(moe_net* n, double target, int* blocks, double* err, void* stream)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_calibrate: NULL net");
    if (!n->finalized) return fail(MOE_ESTATE, "moe_net_calibrate: net is not finalized");
    if (!calibratable(*n)) { if (blocks) *blocks = 0; if (err) *err = 0.0; return MOE_OK; }      // SEDN / lite: no such knob (their AUTO arithmetic has no split-block count)
    if (n->precision != MOE_PREC_MIXED && n->precision != MOE_PREC_FP16X3) return fail(MOE_ESTATE, "moe_net_calibrate: finalize with MOE_PREC_AUTO or MOE_PREC_MIXED first");
    const int rc = calibrate_blocks(*n, target, (hipStream_t)stream);
    if (rc) { n->finalized = false; return rc; }      // (a failure half-way may have left the other arithmetic's weights on the device: finalize again)
    if (blocks) *blocks = n->calib_blocks;
    if (err) *err = n->calib_err;
    return MOE_OK;
}

int moe_net_exact_blocks(const moe_net* n)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_exact_blocks: NULL net");
    return (n->precision == MOE_PREC_MIXED && calibratable(*n)) ? exact_blocks_of(*n) : 0;
}

int64_t moe_net_workspace_bytes(const moe_net* n, int B, int h, int w)
{
    if (!n || B < 1 || h < 1 || w < 1) return fail(MOE_EINVAL, "moe_net_workspace_bytes: bad argument");
    if (!n->finalized) return fail(MOE_ESTATE, "moe_net_workspace_bytes: net is not finalized");
    return (int64_t)workspace_need(*const_cast<moe_net*>(n), B, h, w);
}

static void pipe_destroy(moe_net& n)
{
    for (auto& ps : n.pipe) {
        if (ps.main) { (void)hipStreamSynchronize(ps.main); (void)hipStreamDestroy(ps.main); }
        if (ps.side) { (void)hipStreamSynchronize(ps.side); (void)hipStreamDestroy(ps.side); }
        for (hipEvent_t e : {ps.ev_fork, ps.ev_join, ps.entry, ps.done}) if (e) (void)hipEventDestroy(e);
        if (ps.ws) (void)hipFree(ps.ws);
        ps = moe_net::PipeSet{};
    }
    n.pipe_prev_valid = false;
    n.pipe_last_stream = nullptr;
}

// Small forwards of the SR nets that may run ahead of the caller's stream (the per-tile calls of the reference's loop: up to four planes of 256 x 256)
static bool overlap_eligible(const moe_net& n, int B, int h, int w)
{
    const bool sr = n.arch == MOE_ARCH_NET2X || n.arch == MOE_ARCH_NET3X || n.arch == MOE_ARCH_NET4X;
    return sr && n.finalized && n.opt.overlap_calls && !n.debug && n.opt.conv_impl == 2 && (n.precision == MOE_PREC_MIXED || n.precision == MOE_PREC_FP16) &&
           (long long)B * h * w <= 4ll * 65536 && n.prof_keys.empty() && n.opt.repeat_key.empty();
}

// The reference's tile loop (python/imageProcess.py:164-170) is  r = model(x[..., tile]); blend(r, canvas)  per tile: on ONE stream forward k+1 queues behind the blend of
// tile k, which waits for forward k -- although forward k+1 needs nothing tile k produced.  A 3-plane forward cannot fill 256 CUs (its kernels' workgroups each preload 288
// weight registers; 702 ARSB patches over 256 persistent workgroups: DESIGN.md section 4.10), so the drop-in loop ran at 0.84-0.86 of the device-resident path.  With
// MOE_FWD_INPUT_SINCE_PREV the caller states what makes the overlap legal -- "x was complete on `stream` when the PREVIOUS forward of this net was enqueued" -- and the forward
// runs on one of two internal (stream, workspace) sets behind the previous call's ENTRY event instead of behind everything enqueued since; its last kernel, the only one that
// touches the caller's memory (y), waits for this call's own entry event, and the caller's stream waits for the forward's completion before the call returns: whatever the
// caller enqueues next (the blend) is ordered as before.  Without the flag (or for the first call of a burst) the dependency is this call's entry event: plain stream order.
static int forward_pipelined(moe_net* n, const void* x, int x_dtype, int B, int h, int w, int64_t sB, int64_t sH, int64_t sW, void* y, int y_dtype, hipStream_t s, unsigned flags)
{
    HIP_TRY(hipSetDevice(n->device));
    moe_net::PipeSet& ps = n->pipe[n->pipe_next];
    moe_net::PipeSet& prev = n->pipe[n->pipe_next ^ 1];
    if (!ps.main) {
        HIP_TRY(hipStreamCreateWithFlags(&ps.main, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ps.entry, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(ps.entry, s));
    const bool since_prev = (flags & MOE_FWD_INPUT_SINCE_PREV) && n->pipe_prev_valid && n->pipe_last_stream == s && prev.entry;
    HIP_TRY(hipStreamWaitEvent(ps.main, since_prev ? prev.entry : ps.entry, 0));
    n->pipe_next ^= 1;
    auto swap_set = [&]() { std::swap(n->ws, ps.ws); std::swap(n->ws_bytes, ps.ws_bytes); std::swap(n->side, ps.side); std::swap(n->ev_fork, ps.ev_fork); std::swap(n->ev_join, ps.ev_join); };
    swap_set();
    const int groups0 = n->max_groups;
    n->max_groups = n->opt.overlap_groups > 0 ? std::max(16, std::min(n->opt.overlap_groups, groups0)) : std::max(16, groups0 / 2);
    n->out_gate = since_prev ? ps.entry : nullptr;
    const int fork0 = n->opt.branch_streams;
    if (!n->opt.overlap_fork) n->opt.branch_streams = 0;
    const int rc = forward_dev(*n, x, x_dtype, B, h, w, sB, sH, sW, nullptr, y, y_dtype, nullptr, ps.main, true);
    n->opt.branch_streams = fork0;
    n->out_gate = nullptr;
    n->max_groups = groups0;
    swap_set();
    // the caller's stream continues behind this forward -- also when it failed half-way (kernels may be in flight on the set's streams)
    if (hipEventRecord(ps.done, ps.main) != hipSuccess || hipStreamWaitEvent(s, ps.done, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(ps.main); }
    n->pipe_prev_valid = rc == MOE_OK;
    n->pipe_last_stream = s;
    return rc;
}

int moe_net_forward_ex(moe_net* n, const void* x, int x_dtype, int B, int h, int w, int64_t sB, int64_t sH, int64_t sW,
                       const int64_t* x_off, void* y, int y_dtype, const int64_t* y_off, void* stream, unsigned flags)
{
    if (!n || !x || !y) return fail(MOE_EINVAL, "moe_net_forward: NULL argument");
    if (flags & ~(unsigned)MOE_FWD_INPUT_SINCE_PREV) return fail(MOE_EINVAL, "moe_net_forward_ex: unknown flags 0x%x", flags);
    if ((flags & MOE_FWD_INPUT_SINCE_PREV) && !x_off && !y_off && B >= 1 && h >= 1 && w >= 1 && overlap_eligible(*n, B, h, w)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
            return forward_pipelined(n, x, x_dtype, B, h, w, sB, sH, sW, y, y_dtype, (hipStream_t)stream, flags);
        (void)hipGetLastError();
    }
    n->pipe_prev_valid = false;          // (a forward on the caller's own stream ends a burst)
    return moe_net_forward(n, x, x_dtype, B, h, w, sB, sH, sW, x_off, y, y_dtype, y_off, stream);
}

int moe_net_forward(moe_net* n, const void* x, int x_dtype, int B, int h, int w, int64_t sB, int64_t sH, int64_t sW,
                    const int64_t* x_off, void* y, int y_dtype, const int64_t* y_off, void* stream)
{
    if (!n || !x || !y) return fail(MOE_EINVAL, "moe_net_forward: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    long long* xo = nullptr;
    long long* yo = nullptr;
    moe_net::OffSlot* slot = nullptr;
    if (x_off || y_off) {
        // The host tables ride to the device on the launch stream: a slot of a small ring (pinned host copy + device copy) per call,
        // reused once the event recorded behind the forward that reads it has fired -- no hipMalloc, no blocking copy, no stream synchronisation.
        HIP_TRY(hipSetDevice(n->device >= 0 ? n->device : 0));
        moe_net::OffSlot& sl = n->off_ring[n->off_next];
        n->off_next = (n->off_next + 1) % 4;
        if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));          // (only when four such forwards are still in flight)
        const size_t need = (size_t)B * 2;
        if (need > sl.cap) {
            if (sl.host) { (void)hipHostFree(sl.host); sl.host = nullptr; }
            if (sl.dev) { (void)hipFree(sl.dev); sl.dev = nullptr; }
            sl.cap = 0;
            const size_t cap = std::max<size_t>(need, 256);
            HIP_TRY(hipHostMalloc((void**)&sl.host, cap * 8, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void**)&sl.dev, cap * 8));
            sl.cap = cap;
        }
        if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        for (int i = 0; i < B; ++i) { sl.host[i] = x_off ? x_off[i] : 0; sl.host[B + i] = y_off ? y_off[i] : 0; }
        HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, need * 8, hipMemcpyHostToDevice, s));
        sl.used = true;
        slot = &sl;
        if (x_off) xo = sl.dev;
        if (y_off) yo = sl.dev + B;
    }
    bool mult8 = true;
    if (y_off) for (int i = 0; i < B; ++i) mult8 = mult8 && (y_off[i] % 8 == 0);
    const int rc = forward_dev(*n, x, x_dtype, B, h, w, sB, sH, sW, xo, y, y_dtype, yo, s, mult8);
    // the slot's event covers the copy AND every kernel of this forward that reads the tables: recorded behind them, on their stream, so that a later call on
    // ANOTHER stream cannot overwrite the tables while this forward is still in flight (the host wait above is on this event)
    if (slot && hipEventRecord(slot->done, s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); }
    return rc;
}

int moe_net_set_option(moe_net* n, const char* key, const char* value)
{
    if (!n || !key || !value) return fail(MOE_EINVAL, "moe_net_set_option: NULL argument");
    if (!n->opt.set(key, value)) return fail(MOE_EINVAL, "moe_net_set_option: unknown option or value \"%s\" = \"%s\"", key, value);
    if (n->lut_state != 2) drop_lut(*n);      // (a kernel-form switch may change the U branch's bits: the table is refilled by the next fp16 forward)
    if (!strcmp(key, "max_groups") && n->finalized) n->max_groups = n->opt.max_groups > 0 ? n->opt.max_groups : conv_mfma_max_groups();
    return MOE_OK;
}

int moe_device_info(int device, int64_t info[8])
{
    if (!info) return fail(MOE_EINVAL, "moe_device_info: NULL argument");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    int wall = 0;
    (void)hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, device);
    const int64_t v[8] = {p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize, (int64_t)p.totalGlobalMem, wall,
                          (int64_t)p.maxSharedMemoryPerMultiProcessor};
    memcpy(info, v, sizeof v);
    return MOE_OK;
}

int moe_net_set_profile(moe_net* n, const char* layer_substrings)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_set_profile: NULL net");
    n->prof_keys.clear();
    std::string all = layer_substrings ? layer_substrings : "";
    size_t pos = 0;
    while (pos <= all.size() && !all.empty()) {
        const size_t c = all.find(',', pos);
        const std::string k = all.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        if (!k.empty()) n->prof_keys.push_back(k);
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    n->prof_used = 0;
    return MOE_OK;
}

int moe_net_get_profile_at(moe_net* n, int index, double* total_ms, int64_t* launches, double* flops)
{
    if (!n || !total_ms || !launches || !flops) return fail(MOE_EINVAL, "moe_net_get_profile: NULL argument");
    double ms = 0, fl = 0;
    int64_t cnt = 0;
    for (size_t i = 0; i < n->prof_used; ++i) {
        const moe_net::ProfRec& r = n->prof_ev[i];
        if (r.key != index) continue;
        HIP_TRY(hipEventSynchronize(r.e1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
        ms += t; fl += r.flops; ++cnt;
    }
    *total_ms = ms; *launches = cnt; *flops = fl;
    return MOE_OK;
}

int moe_net_get_profile(moe_net* n, double* total_ms, int64_t* launches, double* flops)
{
    int rc = moe_net_get_profile_at(n, 0, total_ms, launches, flops);
    if (rc == MOE_OK) n->prof_used = 0;
    return rc;
}

int64_t moe_net_max_tile_pixels(const moe_net* n)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_max_tile_pixels: NULL net");
    return (int64_t)(kSpRange / sp_bytes_per_pixel(*n));
}

int moe_net_set_exact_blocks(moe_net* n, int blocks)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_set_exact_blocks: NULL net");
    if (blocks < -1 || blocks > 6) return fail(MOE_EINVAL, "moe_net_set_exact_blocks: %d not in -1..6", blocks);
    n->exact_blocks = blocks;
    return MOE_OK;
}

int moe_net_set_debug(moe_net* n, int enable)
{
    if (!n) return fail(MOE_EINVAL, "moe_net_set_debug: NULL net");
    n->debug = enable != 0;
    return MOE_OK;
}

int64_t moe_net_debug_tap(moe_net* n, const char* tap, float* host, int64_t capacity, int64_t shape[4], void* stream)
{
    if (!n || !tap) return fail(MOE_EINVAL, "moe_net_debug_tap: NULL argument");
    auto it = n->taps.find(tap);
    if (it == n->taps.end() || !it->second.dev) return fail(MOE_EINVAL, "no tap named \"%s\" (enable moe_net_set_debug before the forward)", tap);
    const auto& t = it->second;
    const int64_t nel = t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3];
    if (shape) for (int d = 0; d < 4; ++d) shape[d] = t.shape[d];
    if (!host) return nel;
    if (capacity < nel) return fail(MOE_EINVAL, "moe_net_debug_tap: capacity %lld < %lld", (long long)capacity, (long long)nel);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(host, t.dev, (size_t)nel * 4, hipMemcpyDeviceToHost));
    return nel;
}

// ---- planner ------------------------------------------------------------------------------------------
int moe_plan_create(const int64_t shape[3], double ram, double ram_coef, int pad, int scale, int align, int cropsize, moe_plan** out)
{
    if (!shape || !out) return fail(MOE_EINVAL, "moe_plan_create: NULL argument");
    auto p = std::make_unique<moe_plan>();
    std::string err;
    int rc = build_plan(p->p, shape, ram, ram_coef, pad, scale, align, cropsize, err);
    if (rc) return fail(rc, "%s", err.c_str());
    *out = p.release();
    return MOE_OK;
}

void moe_plan_destroy(moe_plan* p)
{
    if (!p) return;
    for (auto& d : p->p.dev) { if (d->blob) (void)hipFree(d->blob); for (auto& e : d->strip_tabs) (void)hipFree(e.second); }
    for (auto& d : p->p.fdev) if (d->blob) (void)hipFree(d->blob);
    for (auto& c : p->p.custom_off) if (c.dev) (void)hipFree(c.dev);
    if (p->p.pool) (void)hipFree(p->p.pool);
    delete p;
}

int moe_plan_info(const moe_plan* p, int64_t info[12])
{
    if (!p || !info) return fail(MOE_EINVAL, "moe_plan_info: NULL argument");
    const Plan& q = p->p;
    const int64_t v[12] = {(int64_t)q.tiles.size(), q.ah.step, q.aw.step, q.out_h, q.out_w, q.pad_h_to, q.pad_w_to, q.pad_sc,
                           q.tile_h, q.tile_w, q.ah.clip, q.aw.clip};
    memcpy(info, v, sizeof v);
    return MOE_OK;
}

int moe_plan_rows(const moe_plan* p, int32_t* rows)
{
    if (!p || !rows) return fail(MOE_EINVAL, "moe_plan_rows: NULL argument");
    memcpy(rows, p->p.row_tab.data(), p->p.row_tab.size() * 4);
    return MOE_OK;
}

// Seam rows / columns of every tile for the wire format (misc_kernels.hip: wire_kernel): per tile 8 ints (ra0, ra1, rb0, rb1, ca0, ca1, cb0, cb1), tile-local.
// A tile's value is read at full precision inside its OWN blend band and wherever a LATER tile along the axis blends over it; the union of those bands,
// clipped to the tile, is covered with at most two ranges per axis (more than two are merged into the hull of the second and the rest: a superset is safe).
static void axis_seams(const std::vector<int>& tab, int i, int out[4])
{
    const int n = (int)tab.size() / 4;
    const int o = tab[i * 4 + 2], ext = tab[i * 4 + 3];
    std::vector<std::pair<int, int>> iv;
    for (int k = 0; k < n; ++k) {                      // (earlier tiles' bands lie before the tile; they are taken along for the clipped last tile: a superset is safe)
        const int a = std::max(tab[k * 4 + 0] - o, 0), b = std::min(tab[k * 4 + 1] - o, ext);
        if (b > a) iv.push_back({a, b});
    }
    std::sort(iv.begin(), iv.end());
    std::vector<std::pair<int, int>> m;
    for (auto& v : iv) {
        if (!m.empty() && v.first <= m.back().second) m.back().second = std::max(m.back().second, v.second);
        else m.push_back(v);
    }
    while (m.size() > 2) { m[1].second = m.back().second; m.pop_back(); }
    out[0] = out[1] = out[2] = out[3] = 0;
    if (m.size() >= 1) { out[0] = m[0].first; out[1] = m[0].second; out[2] = out[3] = m[0].second; }
    if (m.size() == 2) { out[2] = m[1].first; out[3] = m[1].second; }
}

int moe_plan_seams(const moe_plan* p, int32_t* seams)
{
    if (!p || !seams) return fail(MOE_EINVAL, "moe_plan_seams: NULL argument");
    const Plan& q = p->p;
    for (int i = 0; i < q.ah.step; ++i)
        for (int j = 0; j < q.aw.step; ++j) {
            int r[4], c[4];
            axis_seams(q.row_tab, i, r);
            axis_seams(q.col_tab, j, c);
            int32_t* o = seams + ((size_t)i * q.aw.step + j) * 8;
            for (int e = 0; e < 4; ++e) { o[e] = r[e]; o[4 + e] = c[e]; }
        }
    return MOE_OK;
}

static_assert(sizeof(moe_wire_rec) == sizeof(WireRec) && sizeof(WireRec) == 64, "moe_wire_rec layout");

int64_t moe_wire_words(const moe_wire_rec* rec)
{
    if (!rec) return -1;
    WireRec r;
    memcpy(&r, rec, sizeof r);
    return wire_rec_words(r);
}

static int wire_call(bool pack, float* tiles, void* wire, const moe_wire_rec* recs_dev, int n, int64_t max_elems, void* stream)
{
    if (n < 0 || (n > 0 && (!tiles || !wire || !recs_dev))) return fail(MOE_EINVAL, "moe_wire_%s: bad argument", pack ? "pack" : "unpack");
    launch_wire(pack, tiles, (unsigned*)wire, (const WireRec*)recs_dev, n, max_elems, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "wire kernel launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_wire_pack(const float* tiles_dev, void* wire_dev, const moe_wire_rec* recs_dev, int n, int64_t max_elems, void* stream)
{
    return wire_call(true, (float*)tiles_dev, wire_dev, recs_dev, n, max_elems, stream);
}

int moe_wire_unpack(float* tiles_dev, const void* wire_dev, const moe_wire_rec* recs_dev, int n, int64_t max_elems, void* stream)
{
    return wire_call(false, tiles_dev, (void*)wire_dev, recs_dev, n, max_elems, stream);
}

int moe_plan_tiles(const moe_plan* p, int32_t* tiles)
{
    if (!p || !tiles) return fail(MOE_EINVAL, "moe_plan_tiles: NULL argument");
    for (size_t k = 0; k < p->p.tiles.size(); ++k) {
        const TileRect& t = p->p.tiles[k];
        const int32_t v[8] = {t.top, t.bottom, t.left, t.right, t.top_t, t.left_t, t.bsc, t.rsc};
        memcpy(tiles + k * 8, v, sizeof v);
    }
    return MOE_OK;
}

int moe_plan_ramp(const moe_plan* p, float* ramp)
{
    if (!p || !ramp) return fail(MOE_EINVAL, "moe_plan_ramp: NULL argument");
    memcpy(ramp, p->p.ramp.data(), p->p.ramp.size() * 4);
    return MOE_OK;
}

// ---- stitch / run ------------------------------------------------------------------------------------
int64_t moe_plan_pool_elems(const moe_plan* p, int C)
{
    if (!p || C < 1) return fail(MOE_EINVAL, "moe_plan_pool_elems: bad argument");
    return (int64_t)(p->p.pool_elems_per_plane_set / p->p.C * C);
}

int moe_plan_tile_offsets(const moe_plan* p, int C, int64_t* off)
{
    if (!p || !off || C < 1) return fail(MOE_EINVAL, "moe_plan_tile_offsets: bad argument");
    for (size_t k = 0; k < p->p.tiles.size(); ++k) off[k] = p->p.tile_off[k] / p->p.C * C;
    return MOE_OK;
}

int moe_stitch(const moe_plan* p, int device, const float* tiles_dev, const int64_t* tile_off, int C, void* out, int out_dtype, void* stream)
{
    if (!p || !tiles_dev || !out || C < 1) return fail(MOE_EINVAL, "moe_stitch: bad argument");
    HIP_TRY(hipSetDevice(device));
    PlanDeviceCache* d = nullptr;
    int rc = plan_device_tables(p->p, device, C, 0, 0, 0, 0, 1, &d);
    if (rc) return rc;
    long long* toff = nullptr;
    if (tile_off) {   // caller-defined pool layout (e.g. the receive buffer of dist.py): uploaded once per distinct table, then cached on the plan
        const size_t nt = p->p.tiles.size();
        for (auto& c : p->p.custom_off)
            if (c.device == device && c.host.size() == nt && std::equal(c.host.begin(), c.host.end(), tile_off)) { toff = c.dev; break; }
        if (!toff) {
            if (p->p.custom_off.size() >= 16) {
                HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
                (void)hipFree(p->p.custom_off.front().dev);
                p->p.custom_off.erase(p->p.custom_off.begin());
            }
            HIP_TRY(hipMalloc((void**)&toff, nt * 8));
            HIP_TRY(hipMemcpy(toff, tile_off, nt * 8, hipMemcpyHostToDevice));
            p->p.custom_off.push_back(CustomOffsets{device, std::vector<long long>(tile_off, tile_off + nt), toff});
        }
    }
    StitchArgs a{};
    fill_stitch(p->p, *d, a, tiles_dev, toff ? toff : d->tile_off, C, out, out_dtype);
    launch_stitch(a, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "stitch launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_stitch_dev(const moe_plan* p, int device, const float* tiles_dev, const int64_t* tile_off_dev, int C, void* out, int out_dtype, void* stream)
{
    if (!p || !tiles_dev || !tile_off_dev || !out || C < 1) return fail(MOE_EINVAL, "moe_stitch_dev: bad argument");
    HIP_TRY(hipSetDevice(device));
    PlanDeviceCache* d = nullptr;
    int rc = plan_device_tables(p->p, device, C, 0, 0, 0, 0, 1, &d);
    if (rc) return rc;
    StitchArgs a{};
    fill_stitch(p->p, *d, a, tiles_dev, (const long long*)tile_off_dev, C, out, out_dtype);
    launch_stitch(a, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "stitch launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_stitch_band(const moe_plan* p, int device, const float* tiles_dev, const int64_t* tile_off_dev, int C, void* out, int out_dtype,
                    int row0, int row1, int strip, void* stream)
{
    if (!p || !tiles_dev || !tile_off_dev || !out || C < 1) return fail(MOE_EINVAL, "moe_stitch_band: bad argument");
    const Plan& q = p->p;
    const int nrow = q.ah.step;
    if (row0 < 0 || row1 <= row0 || row1 > nrow) return fail(MOE_EINVAL, "moe_stitch_band: tile rows [%d, %d) of %d", row0, row1, nrow);
    HIP_TRY(hipSetDevice(device));
    PlanDeviceCache* d = nullptr;
    int rc = plan_device_tables(q, device, C, 0, 0, 0, 0, 1, &d);
    if (rc) return rc;
    StitchArgs a{};
    fill_stitch(q, *d, a, tiles_dev, (const long long*)tile_off_dev, C, out, out_dtype);
    a.row_lo = row0;
    a.y0 = q.row_tab[row0 * 4 + 1];                                  // S(row0): first un-blended row of the band's first tile row (0 for row 0)
    a.rows = (row1 < nrow ? q.row_tab[row1 * 4 + 1] : q.out_h) - a.y0;
    if (strip && row1 < nrow) {
        // the band ends with the blend band of tile row row1, rows [first, solid): its tiles are present as STRIPS of exactly those pad_sc rows (C planes of
        // pad_sc x width each): a row table in which that tile row starts at `first` and is pad_sc high addresses them
        int* tab = nullptr;
        for (auto& e : d->strip_tabs) if (e.first == row1) tab = e.second;
        if (!tab) {
            std::vector<int> rt(q.row_tab);
            rt[row1 * 4 + 2] = rt[row1 * 4 + 0];
            rt[row1 * 4 + 3] = rt[row1 * 4 + 1] - rt[row1 * 4 + 0];
            HIP_TRY(hipMalloc((void**)&tab, rt.size() * 4));
            HIP_TRY(hipMemcpy(tab, rt.data(), rt.size() * 4, hipMemcpyHostToDevice));
            d->strip_tabs.push_back({row1, tab});
        }
        a.row_tab = tab;
    }
    launch_stitch(a, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "stitch launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_blend_tile(const void* r, int64_t r_sC, int64_t r_sH, void* canvas, int64_t c_sC, int64_t c_sH, int dtype, int C,
                   int top_sc, int left_sc, int bsc, int rsc, int topT, int leftT, int pad_sc, const void* ramp, void* stream)
{
    if (!r || !canvas || C < 1) return fail(MOE_EINVAL, "moe_blend_tile: NULL argument");
    if (dtype != MOE_F32 && dtype != MOE_F16) return fail(MOE_EINVAL, "moe_blend_tile: dtype must be MOE_F32 or MOE_F16");
    const int rh = bsc - top_sc, rw = rsc - left_sc;
    if (rh < 1 || rw < 1 || top_sc < 0 || left_sc < 0 || pad_sc < 0) return fail(MOE_EINVAL, "moe_blend_tile: empty or negative window (%d..%d, %d..%d)", top_sc, bsc, left_sc, rsc);
    // blend(r, x, lt, pad, dim, ..), python/imageProcess.py:120-131: lt < 0 counts from the end; lt < 1: nothing is blended and the whole extent is assigned
    auto band = [&](int lt, int l, int& first, int& solid) {
        if (lt < 0) lt += l;
        if (lt < 1) { first = solid = 0; return true; }
        first = lt - pad_sc; solid = lt;
        return first >= 0 && lt <= l;
    };
    BlendTileArgs a{};
    if (!band(topT, rh, a.r0, a.lt_h) || !band(leftT, rw, a.c0, a.lt_w))
        return fail(MOE_EINVAL, "moe_blend_tile: blend band outside the window (topT %d, leftT %d, pad_sc %d, window %d x %d)", topT, leftT, pad_sc, rh, rw);
    if ((a.lt_h > a.r0 || a.lt_w > a.c0) && !ramp) return fail(MOE_EINVAL, "moe_blend_tile: ramp is NULL");
    a.r = r; a.canvas = canvas; a.ramp = ramp; a.r_sC = r_sC; a.r_sH = r_sH; a.c_sC = c_sC; a.c_sH = c_sH;
    a.C = C; a.rh = rh; a.rw = rw; a.top_sc = top_sc; a.left_sc = left_sc;
    launch_blend_tile(a, dtype == MOE_F16, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "blend launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_run_plan_ex(moe_net* n, const moe_plan* pl, const void* img, int img_dtype, int64_t sC, int64_t sH, int64_t sW,
                    void* out, int out_dtype, int max_tiles, float* pool, int shard_index, int shard_count, int do_stitch, void* stream)
{
    if (!n || !pl || !img || (do_stitch && !out)) return fail(MOE_EINVAL, "moe_run_plan: NULL argument");
    if (!n->finalized) return fail(MOE_ESTATE, "moe_run_plan: net is not finalized");
    const Plan& p = pl->p;
    if (p.sc != n->scale) return fail(MOE_EINVAL, "moe_run_plan: plan scale %d != net scale %d", p.sc, n->scale);
    if (shard_count < 1) { shard_count = 1; shard_index = 0; }
    if (shard_index < 0 || shard_index >= shard_count) return fail(MOE_EINVAL, "moe_run_plan: shard %d of %d", shard_index, shard_count);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(n->device));
    const int C = p.C;
    PlanDeviceCache* d = nullptr;
    int rc = plan_device_tables(p, n->device, C, sC, sH, sW, shard_index, shard_count, &d);
    if (rc) return rc;
    if (!pool) {
        const size_t pool_need = p.pool_elems_per_plane_set;
        if (pool_need > p.pool_elems) {
            if (p.pool) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(p.pool)); p.pool = nullptr; p.pool_elems = 0; }
            if (hipMalloc((void**)&p.pool, pool_need * 4) != hipSuccess) { (void)hipGetLastError(); return fail(MOE_ENOMEM, "tile pool of %zu bytes does not fit", pool_need * 4); }
            p.pool_elems = pool_need;
        }
        pool = p.pool;
    }
    if (max_tiles <= 0) {
        max_tiles = n->opt.tiles_per_batch > 0 ? n->opt.tiles_per_batch : 32;     // (tiles of 256^2 pixels per launch set: 28.97 / 28.59 / 28.42 / 28.28 / 28.33 ms per 1080p x4 frame with 4 / 8 / 12 / 16 / 24 in round 3: fewer pipeline
                                                                                   // fills per pixel; round 4's kernels, 16 / 20 / 28: 24.32 / 24.20 / 24.09 -- the 28 full tiles of a 1080p frame as ONE launch set; a tile's bits do not depend on it)
    }
    for (size_t gi = 0; gi < p.groups.size(); ++gi) {
        const auto& g = p.groups[gi];
        const int nt = d->group_count[gi];
        if (nt < 1) continue;
        // bigger batches for small tiles: keep roughly max_tiles * 256^2 pixels per launch
        const long long px = (long long)g.th * g.tw;
        const int per = (int)std::max<long long>(1, std::min<long long>(nt, (long long)max_tiles * 65536 / std::max<long long>(px, 1)));
        for (int t0 = 0; t0 < nt; t0 += per) {
            const int cnt = std::min(per, nt - t0);
            const long long slot = (long long)(d->group_first[gi] + t0) * C;
            rc = forward_dev(*n, img, img_dtype, cnt * C, g.th, g.tw, 0, sH, sW, d->x_off + slot, pool, MOE_F32, d->y_off + slot, s, d->y_mult8);
            if (rc) return rc;
        }
    }
    if (do_stitch) {
        StitchArgs a{};
        fill_stitch(p, *d, a, pool, d->tile_off, C, out, out_dtype);
        launch_stitch(a, s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MOE_EHIP, "stitch launch failed: %s", hipGetErrorString(e));
    }
    return MOE_OK;
}

int moe_run_plan_tiles(moe_net* n, const moe_plan* pl, const void* imgs, int img_dtype, int64_t frame_stride,
                       int64_t sC, int64_t sH, int64_t sW, int n_frames, float* dst, const int64_t* tile_dst,
                       int max_tiles, void* stream)
{
    if (!n || !pl || !imgs || !dst || !tile_dst || n_frames < 1) return fail(MOE_EINVAL, "moe_run_plan_tiles: bad argument");
    if (!n->finalized) return fail(MOE_ESTATE, "moe_run_plan_tiles: net is not finalized");
    const Plan& p = pl->p;
    if (p.sc != n->scale) return fail(MOE_EINVAL, "moe_run_plan_tiles: plan scale %d != net scale %d", p.sc, n->scale);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(n->device));
    const int C = p.C;
    const long long nt = (long long)p.tiles.size();
    FramesDeviceCache* d = nullptr;
    for (auto& up : p.fdev) {
        FramesDeviceCache& c = *up;
        if (c.blob && c.device == n->device && c.C == C && c.sC == sC && c.sH == sH && c.sW == sW && c.frame_stride == frame_stride &&
            c.n_frames == n_frames && c.tile_dst.size() == (size_t)(nt * n_frames) &&
            std::equal(c.tile_dst.begin(), c.tile_dst.end(), tile_dst)) { d = &c; break; }
    }
    if (!d) {
        if (p.fdev.size() >= 8) {
            HIP_TRY(hipStreamSynchronize(s));
            if (p.fdev.front()->blob) (void)hipFree(p.fdev.front()->blob);
            p.fdev.erase(p.fdev.begin());
        }
        p.fdev.push_back(std::make_unique<FramesDeviceCache>());
        d = p.fdev.back().get();
        std::vector<long long> xo, yo;
        int slot = 0;
        for (const auto& g : p.groups) {      // same-shaped tiles of ALL frames share launches
            d->group_first.push_back(slot);
            int cnt = 0;
            const long long plane = (long long)(g.th * p.sc) * (g.tw * p.sc);
            for (int f = 0; f < n_frames; ++f)
                for (int k : g.tiles) {
                    const long long at = tile_dst[(long long)f * nt + k];
                    if (at < 0) continue;                                  // not computed by this call
                    const TileRect& t = p.tiles[k];
                    for (int c = 0; c < C; ++c) {
                        xo.push_back((long long)f * frame_stride + (long long)c * sC + (long long)t.top * sH + (long long)t.left * sW);
                        yo.push_back(at + (long long)c * plane);
                    }
                    ++slot; ++cnt;
                }
            d->group_count.push_back(cnt);
        }
        if (xo.empty()) { xo.push_back(0); yo.push_back(0); }
        d->y_mult8 = true;
        for (long long v : yo) d->y_mult8 = d->y_mult8 && (v % 8 == 0);
        HIP_TRY(hipMalloc(&d->blob, xo.size() * 16));
        d->x_off = (long long*)d->blob; d->y_off = d->x_off + xo.size();
        HIP_TRY(hipMemcpy(d->x_off, xo.data(), xo.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->y_off, yo.data(), yo.size() * 8, hipMemcpyHostToDevice));
        d->device = n->device; d->C = C; d->sC = sC; d->sH = sH; d->sW = sW; d->frame_stride = frame_stride;
        d->n_frames = n_frames; d->tile_dst.assign(tile_dst, tile_dst + nt * n_frames);
    }
    if (max_tiles <= 0) {
        max_tiles = n->opt.tiles_per_batch > 0 ? n->opt.tiles_per_batch : 32;     // (as moe_run_plan_ex)
    }
    for (size_t gi = 0; gi < p.groups.size(); ++gi) {
        const auto& g = p.groups[gi];
        const int ntl = d->group_count[gi];
        if (ntl < 1) continue;
        const long long px = (long long)g.th * g.tw;
        const int per = (int)std::max<long long>(1, std::min<long long>(ntl, (long long)max_tiles * 65536 / std::max<long long>(px, 1)));
        for (int t0 = 0; t0 < ntl; t0 += per) {
            const int cnt = std::min(per, ntl - t0);
            const long long slot = (long long)(d->group_first[gi] + t0) * C;
            int rc = forward_dev(*n, imgs, img_dtype, cnt * C, g.th, g.tw, 0, sH, sW, d->x_off + slot, dst, MOE_F32, d->y_off + slot, s, d->y_mult8);
            if (rc) return rc;
        }
    }
    return MOE_OK;
}

int moe_run_plan_frames(moe_net* n, const moe_plan* pl, const void* imgs, int img_dtype, int64_t frame_stride,
                        int64_t sC, int64_t sH, int64_t sW, int n_frames, float* pools, int64_t pool_stride,
                        int owner_index, int owner_count, int max_tiles, void* stream)
{
    if (!n || !pl || !imgs || !pools || n_frames < 1) return fail(MOE_EINVAL, "moe_run_plan_frames: bad argument");
    const Plan& p = pl->p;
    if (owner_count < 1) { owner_count = 1; owner_index = 0; }
    if (owner_index < 0 || owner_index >= owner_count) return fail(MOE_EINVAL, "moe_run_plan_frames: owner %d of %d", owner_index, owner_count);
    if (pool_stride < (int64_t)p.pool_elems_per_plane_set) return fail(MOE_EINVAL, "moe_run_plan_frames: pool stride smaller than one frame's pool");
    const long long nt = (long long)p.tiles.size();
    std::vector<int64_t> at((size_t)(nt * n_frames));
    for (int f = 0; f < n_frames; ++f)
        for (long long k = 0; k < nt; ++k)
            at[(size_t)(f * nt + k)] = ((f * nt + k) % owner_count == owner_index) ? (int64_t)f * pool_stride + p.tile_off[(size_t)k] : -1;
    return moe_run_plan_tiles(n, pl, imgs, img_dtype, frame_stride, sC, sH, sW, n_frames, pools, at.data(), max_tiles, stream);
}

int moe_run_plan(moe_net* n, const moe_plan* pl, const void* img, int img_dtype, int64_t sC, int64_t sH, int64_t sW,
                 void* out, int out_dtype, int max_tiles, void* stream)
{
    return moe_run_plan_ex(n, pl, img, img_dtype, sC, sH, sW, out, out_dtype, max_tiles, nullptr, 0, 1, 1, stream);
}

// ---- image edges ---------------------------------------------------------------------------------------
int moe_to_float(const void* src, int src_dtype, int bits, int H, int W, int C, void* dst, int dst_dtype, int device, void* stream)
{
    if (!src || !dst || H < 1 || W < 1 || C < 1) return fail(MOE_EINVAL, "moe_to_float: bad argument");
    if ((src_dtype != MOE_U8 && src_dtype != MOE_U16) || (dst_dtype != MOE_F32 && dst_dtype != MOE_F16)) return fail(MOE_EINVAL, "moe_to_float: bad dtype");
    HIP_TRY(hipSetDevice(device));
    if (src_dtype == MOE_U8) launch_to_float(src, src_dtype, 255.f, true, H, W, C, dst, dst_dtype, (hipStream_t)stream);
    else launch_to_float(src, src_dtype, 1.f / (float)(1 << bits), false, H, W, C, dst, dst_dtype, (hipStream_t)stream);
    return MOE_OK;
}

int moe_resize(const void* src, void* dst, int dtype, int C, int H, int W, int h, int w, int mode, int device, void* stream)
{
    if (!src || !dst || C < 1 || H < 1 || W < 1 || h < 1 || w < 1) return fail(MOE_EINVAL, "moe_resize: bad argument");
    if (dtype != MOE_F32 && dtype != MOE_F16) return fail(MOE_EINVAL, "moe_resize: dtype must be MOE_F32 or MOE_F16");
    if (mode < MOE_RESIZE_NEAREST || mode > MOE_RESIZE_BICUBIC) return fail(MOE_EINVAL, "moe_resize: unknown mode %d", mode);
    HIP_TRY(hipSetDevice(device));
    launch_resize(src, dst, dtype, C, H, W, h, w, mode, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MOE_EHIP, "resize launch failed: %s", hipGetErrorString(e));
    return MOE_OK;
}

int moe_to_output(const void* src, int src_dtype, int H, int W, int C, int bits, void* dst, int dst_dtype, int device, void* stream)
{
    if (!src || !dst || H < 1 || W < 1 || C < 1 || bits < 1 || bits > 16) return fail(MOE_EINVAL, "moe_to_output: bad argument");
    if ((dst_dtype != MOE_U8 && dst_dtype != MOE_U16) || (src_dtype != MOE_F32 && src_dtype != MOE_F16)) return fail(MOE_EINVAL, "moe_to_output: bad dtype");
    if (dst_dtype == MOE_U8 && bits > 8) return fail(MOE_EINVAL, "moe_to_output: %d bits do not fit MOE_U8", bits);
    HIP_TRY(hipSetDevice(device));
    launch_to_output(src, src_dtype, H, W, C, (float)(1 << bits), dst, dst_dtype, (hipStream_t)stream);
    return MOE_OK;
}

}  // extern "C"
