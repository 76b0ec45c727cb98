// extern "C" entry points of librsu_hip.so (declared in include/rsu.h): argument checks, tile planning,
// kernel-configuration choice and launches. No torch types, no hidden allocations: every buffer is the
// caller's; the only library-owned device memory is a 4-KiB page of zeros used for out-of-window reads.
#include <array>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cstddef>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/rsu.h"
#include "elementwise.h"
#include "igemm.h"

static std::atomic<int> g_last_hip_error{0};
static void rsu_set_hip_error(int e) { g_last_hip_error.store(e); }

__device__ uint4 g_zero_page_dev[256];  // 4 KiB, zero-initialised

// (one copy of the page per device: the address is looked up for the CURRENT device -- the caller's stream must belong to it, see
// rsu.h "devices")
static const void* zero_page() {
    static void* ptr[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ptr[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page_dev)) != hipSuccess) return nullptr;
        ptr[dev] = q;
    }
    return ptr[dev];
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }
static inline unsigned magic32(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
static inline int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

extern "C" const char* rsu_version(void) { return "rsu-hip 0.1 (gfx950)"; }
extern "C" int rsu_last_hip_error(void) { return g_last_hip_error.load(); }
// CUs the persistent conv launches plan for (one workgroup each). Every MFMA entry point takes the number as its `ncu` argument
// (32..256); ncu = 0 means this process-wide DEFAULT (256 = the whole chip; tests lower it to run every op at another budget). The
// host never has to change the default between launches: unet.py passes the share of each stream with the launch.
static std::atomic<int> g_cu_budget{256};
extern "C" int rsu_set_cu_budget(int ncu) {
    if (ncu < 32 || ncu > 256) return RSU_EINVAL;
    g_cu_budget.store(ncu);
    return RSU_OK;
}
extern "C" int rsu_get_cu_budget(void) { return g_cu_budget.load(); }
// the budget of one launch: its own argument, or the default; -1 = invalid
static int launch_ncu(int ncu) {
    if (ncu == 0) return g_cu_budget.load();
    return (ncu < 32 || ncu > 256) ? -1 : ncu;
}
// Tile-shape choice of the persistent conv launches by measurement. Every shape computes bit-identical results (the reduction
// order of an output element does not depend on the tile it lies in; tests/test_gpu_cfg_matrix.py), so the choice only affects
// speed. Modes (rsu_set_autotune): RSU_TUNE_OFF the planner's cost model decides (as does RSU_FWD2_CFG >= 0); RSU_TUNE_LOOKUP (the
// default) a launch uses the measured shape of its (geometry, flags, CU budget) when the table holds one and NEVER measures -- the
// launch entry points stay asynchronous; RSU_TUNE_MEASURE a launch whose key is missing times every admissible shape on an idle
// device (it synchronises the device) and records the fastest: the host switches this on for one explicit, untimed tuning pass
// (UNet.tune) and back to LOOKUP afterwards.
static std::atomic<int> g_autotune{RSU_TUNE_LOOKUP};
static std::mutex g_tune_mutex;
static std::map<std::array<int, 16>, int> g_tuned;
extern "C" int rsu_set_autotune(int mode) {
    if (mode < RSU_TUNE_OFF || mode > RSU_TUNE_MEASURE) return RSU_EINVAL;
    g_autotune.store(mode);
    return RSU_OK;
}
extern "C" int rsu_get_autotune(void) { return g_autotune.load(); }
extern "C" int rsu_autotune_entries(void) {
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    return (int)g_tuned.size();
}

// the measured choices as rows of 17 ints (16 key words + the choice): a profile run imports the table a bench run exported, so that
// its kernel statistics hold no timing launches
extern "C" int rsu_autotune_export(int* rows, int capacity) {
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    int n = 0;
    for (const auto& kv : g_tuned) {
        if (rows && n < capacity) {
            for (int i = 0; i < 16; ++i) rows[n * 17 + i] = kv.first[i];
            rows[n * 17 + 16] = kv.second;
        }
        ++n;
    }
    return n;
}
// rows whose choice word does not name a live tile shape and kernel generation of THIS build (a table exported by another build)
// are skipped; returns the number of rows taken
extern "C" int rsu_autotune_import(const int* rows, int nrows) {
    if (!rows || nrows < 0) return RSU_EINVAL;
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    int taken = 0;
    for (int r = 0; r < nrows; ++r) {
        const int choice = rows[r * 17 + 16], shape = choice & 255, gen = choice >> 8;
        if (choice < 0 || shape >= IGF2_NCFG || igemm_fwd2_cfg_info(shape).TN == 0 || gen < 0 || gen > 1) continue;
        if (gen == 1 && !igemm_pp_has(shape)) continue;
        std::array<int, 16> k;
        for (int i = 0; i < 16; ++i) k[i] = rows[r * 17 + i];
        g_tuned[k] = choice;
        ++taken;
    }
    return taken;
}

extern "C" int rsu_input_size_needed(int output_size, int num_layers, int* input_size) {
    // unet.py:100-115: (L-1) x { assert even; o = (o+4)/2 }, (L-1) x { o = (o+4)*2 }, +4
    if (!input_size || num_layers < 1 || output_size < 1) return RSU_EINVAL;
    long o2 = output_size;  // exact in integers: while o stays even the reference's float maths is integral
    for (int i = 0; i < num_layers - 1; ++i) {
        if (o2 % 2 != 0) return RSU_EINVAL;
        o2 = (o2 + 4) / 2;
    }
    for (int i = 0; i < num_layers - 1; ++i) o2 = (o2 + 4) * 2;
    *input_size = (int)(o2 + 4);
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// tile planning
// ---------------------------------------------------------------------------------------------
// Strip width for a Ho x Wo pixel grid cut into aligned TM-pixel tiles: SW is a power of two dividing TM; a tile is TM/SW full rows
// of a strip; npix_cap = LDS pixels available for the halo tile.
static bool plan_geo_aligned(TileGeo& best, int& lsw_out, int Ho, int Wo, int TM, int kh, int kw, int dil, int stride, int npix_cap,
                             int lsw_mask = 0) {
    if (Ho < 1 || Wo < 1) return false;
    double best_cost = 1e30;
    bool found = false;
    for (int lsw = 3; lsw <= 6; ++lsw) {
        const int SW = 1 << lsw;
        if (SW > TM) break;
        if (lsw_mask && !(lsw_mask & (1 << lsw))) continue;   // (only these strip widths: the launch folds the max-pool into its epilogue)
        const int TR = TM / SW;
        TileGeo g;
        g.SW = SW;
        g.nstrips = cdiv(Wo, SW);
        g.tiles_per_strip = cdiv(Ho, TR);
        const int R = (TR - 1) * stride + (kh - 1) * dil + 1;
        g.CW = rup((SW - 1) * stride + (kw - 1) * dil + 1, 8);
        g.npix_max = rup(R * g.CW, 32);
        if (g.npix_max > npix_cap) continue;
        g.inv_SW = magic32(SW);
        g.inv_CW = magic32(g.CW);
        const double waste = (double)g.nstrips * SW * g.tiles_per_strip * TR / ((double)Ho * Wo);
        const double halo = (double)g.npix_max / TM;
        const double cost = waste * (1.0 + 0.04 * halo);
        if (cost < best_cost) {
            best_cost = cost;
            best = g;
            lsw_out = lsw;
            found = true;
        }
    }
    return found;
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
static int kpad_total(const int* seg_c, int nseg) {
    int k = 0;
    for (int i = 0; i < nseg; ++i) k += rup(seg_c[i], 32);
    return k;
}
extern "C" size_t rsu_packed_bytes(int taps, int rows, const int* seg_c, int nseg) {
    return (size_t)taps * kpad_total(seg_c, nseg) * rup(rows, 128) * 2;
}
static int make_pack(PackParams& pp, int ntap, int rows, const int* seg_c, int nseg, long s_tap, long s_row, long s_k, int flip) {
    if (nseg < 1 || nseg > 3) return RSU_EINVAL;
    pp.nchunks = kpad_total(seg_c, nseg) / 32;
    pp.ntap = ntap;
    pp.ntiles = rup(rows, 128) / 16;
    pp.rows = rows;
    pp.nseg = nseg;
    for (int i = 0; i < 3; ++i) pp.seg_c[i] = i < nseg ? seg_c[i] : 0;
    pp.s_tap = s_tap;
    pp.s_row = s_row;
    pp.s_k = s_k;
    pp.flip = flip;
    return RSU_OK;
}
static int do_pack(const float* src, void* dst, int ntap, int rows, const int* seg_c, int nseg, long s_tap, long s_row, long s_k,
                   int flip, hipStream_t st) {
    if (!src || !dst || nseg < 1 || nseg > 3) return RSU_EINVAL;
    PackParams pp;
    pp.nchunks = kpad_total(seg_c, nseg) / 32;
    pp.ntap = ntap;
    pp.ntiles = rup(rows, 128) / 16;
    pp.rows = rows;
    pp.nseg = nseg;
    for (int i = 0; i < 3; ++i) pp.seg_c[i] = i < nseg ? seg_c[i] : 0;
    pp.s_tap = s_tap;
    pp.s_row = s_row;
    pp.s_k = s_k;
    pp.flip = flip;
    HIP_CHECK_RET(ew_pack(src, dst, pp, st));
    return RSU_OK;
}
extern "C" int rsu_pack_conv_fwd(const float* w, void* packed, int k, int Cin, int Cout, const int* seg_c, int nseg, rsu_stream_t stream) {
    int one[1] = {Cin};
    if (!seg_c) { seg_c = one; nseg = 1; }
    int sum = 0;
    for (int i = 0; i < nseg; ++i) sum += seg_c[i];
    if (sum != Cin || (k != 3 && k != 1)) return RSU_EINVAL;
    // HWIO: element (tap, ci, co) at (tap*Cin + ci)*Cout + co; rows = co, k = ci
    return do_pack(w, packed, k * k, Cout, seg_c, nseg, (long)Cin * Cout, 1, Cout, 0, (hipStream_t)stream);
}
extern "C" int rsu_pack_conv_bwd(const float* w, void* packed, int k, int Cin_total, int ci_off, int ci_cnt, int Cout,
                                 rsu_stream_t stream) {
    int seg[1] = {Cout};
    if (!w || ci_off < 0 || ci_cnt < 1 || ci_off + ci_cnt > Cin_total) return RSU_EINVAL;
    // rows = ci (of the slice), k = co, taps flipped
    return do_pack(w + (long)ci_off * Cout, packed, k * k, ci_cnt, seg, 1, (long)Cin_total * Cout, Cout, 1, 1, (hipStream_t)stream);
}
extern "C" int rsu_pack_convT_fwd(const float* K, void* packed, int Cin, int Cout, rsu_stream_t stream) {
    int seg[1] = {Cin};
    // K [a][b][co][ci]: four single-tap matrices, laid out back to back (one per output phase)
    const size_t per = rsu_packed_bytes(1, Cout, seg, 1);
    for (int ab = 0; ab < 4; ++ab) {
        int rc = do_pack(K + (long)ab * Cout * Cin, (char*)packed + ab * per, 1, Cout, seg, 1, 0, Cin, 1, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return RSU_OK;
}
extern "C" int rsu_pack_convT_bwd(const float* K, void* packed, int Cin, int Cout, rsu_stream_t stream) {
    int seg[1] = {Cout};
    // rows = ci, k = co, 4 taps (a,b) unflipped
    return do_pack(K, packed, 4, Cin, seg, 1, (long)Cout * Cin, 1, Cin, 0, (hipStream_t)stream);
}

// ---- batched packing table
extern "C" size_t rsu_pack_table_entry_bytes(void) { return sizeof(PackJob); }
extern "C" int rsu_pack_table_add(void* host_table, int index, int kind, const float* w, void* packed, int k, int Cin_total, int ci_off,
                                  int ci_cnt, int Cout, const int* seg_c, int nseg) {
    if (!host_table || !w || !packed || index < 0) return RSU_EINVAL;
    PackJob* t = (PackJob*)host_table + index;
    int one[1];
    int rc = RSU_EINVAL, used = 1;
    memset(t, 0, sizeof(PackJob));
    t->src = w;
    t->dst = (bf16_t*)packed;
    switch (kind) {
        case RSU_PACK_CONV_FWD:
            one[0] = Cin_total;
            if (!seg_c) { seg_c = one; nseg = 1; }
            rc = make_pack(t->pp, k * k, Cout, seg_c, nseg, (long)Cin_total * Cout, 1, Cout, 0);
            break;
        case RSU_PACK_CONV_BWD:
            one[0] = Cout;
            t->src = w + (long)ci_off * Cout;
            rc = make_pack(t->pp, k * k, ci_cnt, one, 1, (long)Cin_total * Cout, Cout, 1, 1);
            break;
        case RSU_PACK_CONVT_FWD: {
            one[0] = Cin_total;
            const size_t per = rsu_packed_bytes(1, Cout, one, 1);
            for (int ab = 0; ab < 4; ++ab) {
                memset(t + ab, 0, sizeof(PackJob));
                t[ab].src = w + (long)ab * Cout * Cin_total;
                t[ab].dst = (bf16_t*)((char*)packed + ab * per);
                rc = make_pack(t[ab].pp, 1, Cout, one, 1, 0, Cin_total, 1, 0);
            }
            used = 4;
            break;
        }
        case RSU_PACK_CONVT_BWD:
            one[0] = Cout;
            rc = make_pack(t->pp, 4, Cin_total, one, 1, (long)Cout * Cin_total, 1, Cin_total, 0);
            break;
        case RSU_PACK_CONV_FIRST:
            one[0] = 3;
            rc = make_pack(t->pp, 9, Cout, one, 1, (long)3 * Cout, 1, Cout, 0);
            break;
    }
    return rc == RSU_OK ? used : rc;
}
extern "C" int rsu_pack_table_finish(void* host_table, int nentries, int* total_blocks) {
    if (!host_table || nentries < 1 || !total_blocks) return RSU_EINVAL;
    PackJob* t = (PackJob*)host_table;
    int b = 0;
    for (int i = 0; i < nentries; ++i) {
        t[i].block_start = b;
        b += ew_pack_blocks(t[i].pp);
    }
    *total_blocks = b;
    return RSU_OK;
}
extern "C" int rsu_pack_table_run(const void* dev_table, int nentries, int total_blocks, rsu_stream_t stream) {
    if (!dev_table || nentries < 1 || total_blocks < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_pack_many((const PackJob*)dev_table, nentries, total_blocks, (hipStream_t)stream));
    return RSU_OK;
}

// ---- Momentum + re-pack in one pass (k_update_pack_many): the table of a network's variables
extern "C" size_t rsu_update_table_entry_bytes(void) { return sizeof(UpJob); }
extern "C" int rsu_update_table_add_plain(void* host_table, int index, float* w, float* acc, const float* g, long n) {
    if (!host_table || index < 0 || !w || !acc || !g || n < 1) return RSU_EINVAL;
    if (((uintptr_t)w | (uintptr_t)acc | (uintptr_t)g) & 15) return RSU_EINVAL;
    UpJob* t = (UpJob*)host_table + index;
    memset(t, 0, sizeof(UpJob));
    t->w = w; t->acc = acc; t->g = g; t->n = n; t->kind = 0;
    return 1;
}
extern "C" int rsu_update_table_add(void* host_table, int index, int kind, float* w, float* acc, const float* g, void* packed_fwd,
                                    void* const* packed_bwd, int Cin_total, int Cout, const int* seg_c, int nseg) {
    if (!host_table || index < 0 || !w || !acc || !g || !packed_fwd || Cout % 8 || Cout < 8) return RSU_EINVAL;
    if (((uintptr_t)w | (uintptr_t)acc | (uintptr_t)g) & 15) return RSU_EINVAL;
    UpJob* t = (UpJob*)host_table + index;
    memset(t, 0, sizeof(UpJob));
    t->w = w; t->acc = acc; t->g = g; t->kind = 1;
    int one[1] = {Cin_total};
    if (!seg_c) { seg_c = one; nseg = 1; }
    if (nseg < 1 || nseg > 3) return RSU_EINVAL;
    auto set_segs = [&](const int* sc, int ns) {   // segments of R1
        t->nseg = ns;
        int r0 = 0, b0 = 0;
        for (int i = 0; i < ns; ++i) {
            t->seg_c[i] = sc[i]; t->seg_r0[i] = r0; t->seg_blk0[i] = b0;
            r0 += sc[i];
            b0 += cdiv(sc[i], 32);
        }
        t->nrb = b0;
        return r0;
    };
    switch (kind) {
        case RSU_PACK_CONV_FWD: {   // HWIO [9][Cin][Cout]: forward pack (rows co, k ci over the 32-padded sources) + one backward-data pack per source
            if (set_segs(seg_c, nseg) != Cin_total) return RSU_EINVAL;
            t->ntap = 9; t->R1 = Cin_total; t->R2 = Cout; t->ncb = cdiv(Cout, 32);
            UpDest& f = t->d[0];
            f.base[0] = (bf16_t*)packed_fwd; f.orient = 0; f.ntap = 9; f.tapmode = 0; f.ntiles[0] = rup(Cout, 128) / 16;
            int c0 = 0;
            for (int i = 0; i < nseg; ++i) { f.chunk0[i] = c0; c0 += cdiv(seg_c[i], 32); }
            t->ndest = 1;
            if (packed_bwd) {
                UpDest& b = t->d[1];
                b.orient = 1; b.ntap = 9; b.tapmode = 1;
                for (int i = 0; i < nseg; ++i) {
                    if (!packed_bwd[i]) return RSU_EINVAL;
                    b.base[i] = (bf16_t*)packed_bwd[i];
                    b.ntiles[i] = rup(seg_c[i], 128) / 16;
                }
                t->ndest = 2;
            }
            break;
        }
        case RSU_PACK_CONVT_FWD: {   // K [2][2][Cout][Cin]: four forward matrices (rows co, k ci) + the backward-data pack (rows ci, k co, 4 taps)
            if (Cin_total % 8) return RSU_EINVAL;
            int sc[1] = {Cout};
            set_segs(sc, 1);
            t->ntap = 4; t->R1 = Cout; t->R2 = Cin_total; t->ncb = cdiv(Cin_total, 32);
            UpDest& f = t->d[0];
            int sk[1] = {Cin_total};
            f.base[0] = (bf16_t*)packed_fwd; f.orient = 1; f.ntap = 1; f.tapmode = 2; f.ntiles[0] = rup(Cout, 128) / 16;
            f.tap_buf_stride = (long)(rsu_packed_bytes(1, Cout, sk, 1) / 2);
            t->ndest = 1;
            if (packed_bwd) {
                if (!packed_bwd[0]) return RSU_EINVAL;
                UpDest& b = t->d[1];
                b.base[0] = (bf16_t*)packed_bwd[0]; b.orient = 0; b.ntap = 4; b.tapmode = 0; b.ntiles[0] = rup(Cin_total, 128) / 16; b.chunk0[0] = 0;
                t->ndest = 2;
            }
            break;
        }
        case RSU_PACK_CONV_FIRST: {   // HWIO [9][3][Cout] over the 16-channel input tensor: forward pack only
            int sc[1] = {3};
            set_segs(sc, 1);
            t->ntap = 9; t->R1 = 3; t->R2 = Cout; t->ncb = cdiv(Cout, 32);
            UpDest& f = t->d[0];
            f.base[0] = (bf16_t*)packed_fwd; f.orient = 0; f.ntap = 9; f.tapmode = 0; f.ntiles[0] = rup(Cout, 128) / 16; f.chunk0[0] = 0;
            t->ndest = 1;
            break;
        }
        default: return RSU_EINVAL;
    }
    return 1;
}
extern "C" int rsu_update_table_finish(void* host_table, int nentries, int* total_blocks) {
    if (!host_table || nentries < 1 || !total_blocks) return RSU_EINVAL;
    UpJob* t = (UpJob*)host_table;
    int b = 0;
    for (int i = 0; i < nentries; ++i) {
        t[i].block_start = b;
        b += ew_update_job_blocks(t[i]);
    }
    *total_blocks = b;
    return RSU_OK;
}
extern "C" int rsu_update_table_run(const void* dev_table, int nentries, int total_blocks, float lr, float mu, float gscale, rsu_stream_t stream) {
    if (!dev_table || nentries < 1 || total_blocks < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_update_pack_many((const UpJob*)dev_table, nentries, total_blocks, lr, mu, gscale, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// VALU head / tail
// ---------------------------------------------------------------------------------------------
static bool keep_ok(float keep) { return keep > 0.f && keep <= 1.f; }
extern "C" int rsu_color_adjust_fwd(const float* x, const float* w, const float* b, void* out16, long npix, float keep, unsigned key,
                                    rsu_stream_t stream) {
    if (!x || !w || !b || !out16 || npix < 1 || 3 * npix > 0xffffffffL || !keep_ok(keep)) return RSU_EINVAL;
    HIP_CHECK_RET(ew_color_adjust(x, w, b, out16, npix, keep, key, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_dropout_fwd(const void* x, void* y, long n, float keep, unsigned key, rsu_stream_t stream) {
    if (!x || !y || n < 8 || n % 8 || n > 0xffffffffL || !keep_ok(keep)) return RSU_EINVAL;
    HIP_CHECK_RET(ew_dropout(x, y, n, keep, key, (hipStream_t)stream));
    return RSU_OK;
}
// ---------------------------------------------------------------------------------------------
// igemm_fwd family
// ---------------------------------------------------------------------------------------------
// second-generation kernel: persistent workgroups (one per CU); pick the tile shape with the smallest estimated time
struct Fwd2Plan { int cfg; TileGeo g; int ncob, grid_x, lsw; double cost; int ksplit, cob_group; };
// How many (channel block, reduction slice) units of one pixel tile sit on neighbouring workgroup ids (IgFwdParams::cob_group): the 8 XCDs
// each run a contiguous range of ids through their own L2, so the estimate is the bytes an XCD's run pulls in -- its distinct weight
// slices plus the halo chunks of its distinct (pixel tile, slice) pairs -- and the divisor of the unit count with the smallest sum wins.
// Layers with few pixels and many channels (the deep levels) end up with one or two weight slices per XCD, the full-resolution layers
// with all channel blocks of a tile together as in rounds 1-3.
static int pick_cob_group(int grid, int ncob, int ksplit, int TN, int ktot, int npix_halo, int nchunks) {
    const int force = env_int("RSU_COB_GROUP", -1);
    const int nck = ncob * ksplit;
    if (force >= 0) return (force > 0 && nck % force == 0) ? force : 0;
    const int tstride = grid / nck;
    if (tstride < 1) return 0;
    const double q = grid / 8.0;
    const double w_unit = (double)TN * ktot * 2.0 / ksplit, h_pair = (double)npix_halo * 64.0 * nchunks / ksplit;
    double best = 1e300;
    int best_a = 0;
    for (int a = 1; a <= nck; ++a) {
        if (nck % a || (a > ncob && a % ncob) || (a < ncob && ncob % a)) continue;
        const double groups = std::max(1.0, std::ceil(q / ((double)a * tstride)));
        const double units = std::min((double)nck, a * groups);
        const double tiles = std::min((double)tstride, std::ceil(q / a));
        const double slices = std::max(1.0, std::ceil(units / ncob));
        const double fill = units * w_unit + tiles * slices * h_pair;
        if (fill < best * 0.98) {   // (ties: the smaller group, i.e. fewer distinct weight slices)
            best = fill;
            best_a = a;
        }
    }
    return best_a == nck ? 0 : best_a;
}
static bool plan_fwd2(Fwd2Plan& best, int ncu, int N, int Ho, int Wo, int Cout, int ntap, int kh, int kw, int dil, int stride, int gy, int ktot,
                      int force_cfg, bool shared_chip = false, bool pool = false) {
    double best_cost = 1e300;
    bool have = false;
    // pass 0: channel-block width matched to Cout; pass 1 (only when no such shape fits its halo tile into LDS -- the 2x2 stride-2
    // taps of a narrow transposed conv need 4 x TM halo pixels in a four-slot ring): any width
    for (int pass = 0; pass < 2 && !have; ++pass)
    for (int cfg = 0; cfg < IGF2_NCFG; ++cfg) {
        if (force_cfg >= 0 && cfg != force_cfg) continue;
        if (igemm_fwd2_cfg_info(cfg).TN == 0) continue;  // retired ids
        if (force_cfg < 0 && (cfg == IGF2_CFG_128x320 || cfg == IGF2_CFG_64x640) && !env_int("RSU_CFG_320", 1)) continue;
        const IgFwdCfgInfo ci = igemm_fwd2_cfg_info(cfg);
        if (pass == 0 && Cout <= 64 && ci.TN > 64 && force_cfg < 0) continue;
        if (pass == 0 && Cout > 64 && ci.TN <= 64 && force_cfg < 0) continue;
        const int lsw_mask = pool ? igemm_pp_pool_lsw_mask(cfg) : 0;
        if (pool && !lsw_mask) continue;
        const long fixed = (long)igemm_fwd2_lds_bytes(cfg, ntap, 0);
        const long per_pix = (long)igemm_fwd2_lds_bytes(cfg, ntap, 1) - fixed;  // 64 bytes x halo ring slots
        int cap = (int)((160 * 1024 - fixed) / per_pix);
        const int cap2 = igemm_fwd2_max_pieces(cfg, ntap) * 16;
        if (cap2 < cap) cap = cap2;
        TileGeo g;
        int lsw = 0;
        if (!plan_geo_aligned(g, lsw, Ho, Wo, ci.TM, kh, kw, dil, stride, cap, lsw_mask)) continue;
        const int ncob = cdiv(Cout, ci.TN);
        const long ntile_m = (long)N * g.nstrips * g.tiles_per_strip;
        long workers = ncu / (ncob * gy);  // blockIdx.y slices (transposed-conv phases) share the chip
        if (workers < 1) workers = 1;
        if (workers > ntile_m) workers = ntile_m;
        const long rounds = (ntile_m + workers - 1) / workers;
        // per tile: MFMA work ~ TM*TN per (chunk, tap) + a fixed epilogue/bubble term, which weighs more the shorter the
        // reduction (ktot = taps x channels) is
        double ovh = 4096.0 * 1152.0 / (double)(ktot > 0 ? ktot : 1152);
        if (ovh < 4096.0) ovh = 4096.0;
        if (ovh > 65536.0) ovh = 65536.0;
        // shared_chip: the launch runs beside another stream's persistent kernel (backward-data beside the weight gradients), which
        // fills the idle CUs of a partial last round -- price the rounds fractionally and let the more efficient big tiles win
        const double nrounds = shared_chip ? (double)ntile_m / (double)workers : (double)rounds;
        // measured per-MAC cost of the shapes relative to 128x256: two pixel fragments per wave read 6 LDS fragments per 8 MFMAs
        // and stream twice the weight pieces per MFMA (x1.5 priced: it also covers their shorter stages); five fragments per wave are a little cheaper (x0.96)
        const int ptw = ci.TM / 16 / (ci.threads / 64 / (ci.TN / 64));
        const double eff = ptw <= 2 ? (double)env_int("RSU_PLAN_PT2_PCT", 150) / 100.0
                                    : (ptw == 3 ? (double)env_int("RSU_PLAN_PT3_PCT", 120) / 100.0 : (ptw >= 5 ? 0.96 : 1.0));
        const double cost = nrounds * ((double)ci.TM * ci.TN * eff + ovh);
        const int nchunks = ktot / (32 * ntap);
        if (cost < best_cost) {
            best_cost = cost;
            best.cfg = cfg;
            best.g = g;
            best.ncob = ncob;
            best.grid_x = (int)(workers * ncob);
            best.lsw = lsw;
            best.cost = cost;
            best.ksplit = 1;
            best.cob_group = (ntap == 9 && stride == 1 && dil == 1 && gy == 1) ? pick_cob_group(best.grid_x, ncob, 1, ci.TN, ktot, g.npix_max, nchunks) : 0;
            have = true;
        }
    }
    return have;
}

// (the ping-pong kernel of the launch's dilation: igemm_pp.hip / igemm_pp_d2.hip)
static bool pp_supports(int cfg, const IgFwdParams& p) { return p.dil == 2 ? igemm_pp_d2_supports(cfg, p) : igemm_pp_supports(cfg, p); }
// Split-K plan of a 3x3 stride-1 launch, or false. A PURE FUNCTION OF THE LAYER'S GEOMETRY (never of the CU budget, the tuning table or
// a measurement), so that the same layer sums its reduction in the same order in every schedule, on every box: results stay repeatable bit
// for bit across one / two streams and tuned / untuned runs. The geometry INCLUDES THE BATCH: P counts the tiles of all N images, so the same
// layer is cut into more slices at N = 1 than at N = 4, and an image's deep-layer sums associate differently with its batch size (and between
// the shared-window and the tile-by-tile inference paths). Deciding the slice count from ONE image's tiles was measured (round 5,
// profiles/r05/abenv_perimg_c2.txt / _c4.txt): 8 slices of every N = 4 launch cost the c2 step 14 % and the c4 step 12 % (four times the
// fp32 partial sums), so the batch stays in the rule; tests/test_gpu_ops.py pins the |N = 1 - N = 4| distance per layer instead.
// Rule: with P = (pixel tiles) x (128-channel blocks) of the 128x256 shape, or
// failing that of the 128x128 shape, a layer with P <= 128 cuts its reduction into S = min(256 / P, chunks / 4, 16) >= 2 slices of at least
// four 32-channel chunks; the launch then has P x S workgroups of ONE tile slice each (the hardware deals them over the CUs; a launch
// planned for half the chip simply takes two turns), followed by the finish launch. Needs S x P x (tile floats) of workspace.
static bool plan_split(Fwd2Plan& pl, int N, int Ho, int Wo, int Cout, int ktot, size_t kws_floats, int dil = 1) {
    if (kws_floats == 0 || Cout < 128 || env_int("RSU_KSPLIT", 1) == 0 || env_int("RSU_FWD_GEN", 3) < 3 || env_int("RSU_FWD2_CFG", -1) >= 0) return false;
    const int nchunks = ktot / (32 * 9);
    const int cands[2] = {IGF2_CFG_128x256, IGF2_CFG_128x128};
    for (int k = 0; k < 2; ++k) {
        const int cfg = cands[k];
        const IgFwdCfgInfo ci = igemm_fwd2_cfg_info(cfg);
        const long fixed = (long)igemm_fwd2_lds_bytes(cfg, 9, 0);
        const long per_pix = (long)igemm_fwd2_lds_bytes(cfg, 9, 1) - fixed;
        int cap = (int)((160 * 1024 - fixed) / per_pix);
        const int cap2 = igemm_fwd2_max_pieces(cfg, 9) * 16;
        if (cap2 < cap) cap = cap2;
        TileGeo g;
        int lsw = 0;
        if (!plan_geo_aligned(g, lsw, Ho, Wo, ci.TM, 3, 3, dil, 1, cap)) continue;
        const int ncob = cdiv(Cout, ci.TN);
        const long P = (long)N * g.nstrips * g.tiles_per_strip * ncob;
        if (P > 128) continue;
        int S = (int)(256 / P);
        if (S > nchunks / 4) S = nchunks / 4;
        if (S > 16) S = 16;
        if (S < 2) continue;
        if (k == 0 && P * S < 192 && nchunks / 4 > S) continue;   // (the smaller tiles fill the chip better: see whether they split too)
        if ((size_t)S * (size_t)P * ci.TM * ci.TN > kws_floats) continue;
        {   // ... and the ping-pong kernel must be instantiated for this geometry
            IgFwdParams t;
            memset(&t, 0, sizeof(t));
            t.stride = t.ostride = 1; t.dil = dil; t.lsw = lsw; t.g = g; t.ksplit = S; t.kslab = (float*)16;
            if (!pp_supports(cfg, t)) continue;
        }
        pl.cfg = cfg; pl.g = g; pl.ncob = ncob; pl.lsw = lsw; pl.ksplit = S;
        pl.grid_x = (int)(P * S);
        pl.cost = ((double)ci.TM * ci.TN * (k ? 1.5 : 1.0) / S + 4096.0) * (double)cdiv((int)(P * S), 256);
        pl.cob_group = pick_cob_group(pl.grid_x, ncob, S, ci.TN, ktot, g.npix_max, nchunks);
        return true;
    }
    return false;
}

// persistent conv launch of the chosen generation: the ping-pong kernel (igemm_pp.hip) runs the 3x3 stride-1 launches unless
// RSU_FWD_GEN=2 asks for igemm_fwd2 (same tile shapes, same bits)
// (pp: 0 = igemm_fwd2, 1 = igemm_pp)
static hipError_t launch_persistent(int pp, int cfg, int ntap, const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    if (pp && pp_supports(cfg, p)) {
        hipError_t e = p.dil == 2 ? igemm_pp_d2_launch(cfg, p, gx, st) : igemm_pp_launch(cfg, p, gx, st);
        if (e == hipSuccess && p.ksplit > 1) e = igemm_pp_finish_launch(cfg, p, st);   // split-K: the slices' partial sums -> bf16 output
        return e;
    }
    if (p.ksplit > 1) return hipErrorInvalidValue;   // (only the ping-pong kernel splits its reduction)
    return igemm_fwd2_launch(cfg, ntap, p, gx, gy, st);
}

static int run_fwd(const rsu_src_t* srcs, int nsrc, const void* wp, long wp_y_stride, int ntiles_w, int tile_off, const float* bias,
                   void* out, const void* mask_src, int N, int Hin, int Win, int Ho, int Wo, int Cout, int outC, int ntap, int kw,
                   int dil, int stride, int pad, int oH, int oW, int ostride, int gy, int relu, int accumulate, int ncu_arg, hipStream_t st,
                   void* pool_out = nullptr, void* pool_code = nullptr, float* kws = nullptr, size_t kws_floats = 0) {
    const int kh = ntap / kw;
    const bool pool = pool_out != nullptr;
    const int ncu = launch_ncu(ncu_arg);
    if (ncu < 0) return RSU_EINVAL;
    const long out_bytes = (long)N * oH * oW * outC * 2;
    const int gen = env_int("RSU_FWD_GEN", 3);
    // the kernels address every tensor through 32-bit byte offsets of a buffer descriptor: a tensor must stay below 2 GiB (split
    // the batch otherwise -- L = 6, P = 388 reaches that at 29 patches per call)
    if (out_bytes >= 0x7ffffff0L) return RSU_E2BIG;
    // RSU_FWD_GEN: 2 = igemm_fwd2 only; 3 (default) = the ping-pong kernel where it measured (or, untuned, is expected to be) faster;
    // 4 = the ping-pong kernel wherever it is built (tests)
    const bool pp_ok = gen >= 3 && ntap == 9 && stride == 1 && ostride == 1 && gy == 1 && (dil == 1 || (dil == 2 && env_int("RSU_PP_DIL2", 1) != 0));
    for (int i = 0; i < nsrc; ++i)
        if ((long)N * srcs[i].H * srcs[i].W * srcs[i].C * 2 >= 0x7ffffff0L) return RSU_E2BIG;
    Fwd2Plan pl2;
    int ktot = 0;
    for (int i = 0; i < nsrc; ++i) ktot += rup(srcs[i].C, 32) * ntap;
    const int env_cfg = env_int("RSU_FWD2_CFG", -1);
    const bool shared_chip = pad > 0 && env_int("RSU_PLAN_SHARED", 0) != 0;
    // measured tile-shape choice (see g_tuned): look the launch up, or -- first time -- mark it for tuning below
    const int tune_mode = env_int("RSU_AUTOTUNE", 1) != 0 ? g_autotune.load() : RSU_TUNE_OFF;
    const bool tunable = env_cfg < 0 && !accumulate && tune_mode != RSU_TUNE_OFF;
    std::array<int, 16> tkey = {N, Ho, Wo, Cout, outC, ntap, kw, dil, stride, pad, gy, ktot, nsrc,
                                (mask_src ? 1 : 0) | (relu ? 2 : 0) | (bias ? 4 : 0) | (pool ? 8 : 0) | (pool_code ? 16 : 0), ostride, ncu * 8 + gen};
    int tuned_cfg = -1, tuned_pp = -1;  // the tuned entry holds shape + 256 * (ping-pong kernel)
    if (!kws || accumulate) kws_floats = 0;
    // split-K: decided from the geometry alone (plan_split), outside the tuning table
    Fwd2Plan pls;
    const bool split = pp_ok && !pool && kws_floats > 0 && plan_split(pls, N, Ho, Wo, Cout, ktot, kws_floats, dil);
    bool tune_now = false;
    if (tunable && !split) {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        auto it = g_tuned.find(tkey);
        if (it != g_tuned.end()) {
            tuned_cfg = it->second & 255;
            tuned_pp = it->second >> 8;
        } else {
            tune_now = tune_mode == RSU_TUNE_MEASURE;
        }
    }
    if (split) {
        pl2 = pls;
    } else if (!plan_fwd2(pl2, ncu, N, Ho, Wo, Cout, ntap, kh, kw, dil, stride, gy, ktot, tuned_cfg >= 0 ? tuned_cfg : env_cfg, shared_chip, pool)) {
        // a shape forced through RSU_FWD2_CFG, or a tuned shape (an imported table), whose halo tile does not fit this geometry: plan
        // freely instead (and forget the table entry)
        if (tuned_cfg >= 0) {
            std::lock_guard<std::mutex> lk(g_tune_mutex);
            g_tuned.erase(tkey);
            tuned_pp = -1;
        }
        if (!((env_cfg >= 0 || tuned_cfg >= 0) && plan_fwd2(pl2, ncu, N, Ho, Wo, Cout, ntap, kh, kw, dil, stride, gy, ktot, -1, shared_chip, pool)))
            return RSU_EINVAL;
    }
    IgFwdParams p;
    memset(&p, 0, sizeof(p));
    p.nsrc = nsrc;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].ptr || srcs[i].C % 8) return RSU_EINVAL;
        if (srcs[i].oy < 0 || srcs[i].ox < 0 || srcs[i].oy + Hin > srcs[i].H || srcs[i].ox + Win > srcs[i].W) return RSU_EINVAL;
        p.src[i].ptr = (const bf16_t*)srcs[i].ptr;
        p.src[i].H = srcs[i].H;
        p.src[i].W = srcs[i].W;
        p.src[i].C = srcs[i].C;
        p.src[i].oy = srcs[i].oy;
        p.src[i].ox = srcs[i].ox;
        p.nchunk[i] = cdiv(srcs[i].C, 32);
    }
    p.wp = (const bf16_t*)wp;
    p.wp_y_stride = wp_y_stride;
    p.ntiles_w = ntiles_w;
    p.tile_off = tile_off;
    p.bias = bias;
    p.out = (bf16_t*)out;
    p.mask_src = (const bf16_t*)mask_src;
    p.pool_out = (bf16_t*)pool_out;
    p.pool_code = (unsigned char*)pool_code;
    p.zero_page = zero_page();
    if (!p.zero_page) return RSU_EHIP;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Ho = Ho; p.Wo = Wo;
    p.Cout = Cout; p.outC = outC;
    p.dil = dil; p.stride = stride; p.pad = pad;
    p.oH = oH; p.oW = oW; p.ostride = ostride;
    p.relu = relu; p.accumulate = accumulate;
    {
        // kernel generation: the ping-pong kernel wherever it is instantiated (measured 10-20 % faster than igemm_fwd2 at every
        // shape; launch_persistent falls back to igemm_fwd2 for the rest) unless the tuner measured otherwise for this geometry
        const int pp = (pp_ok && !accumulate && igemm_pp_has(pl2.cfg) && (gen >= 4 || tuned_pp != 0 || pool || pl2.ksplit > 1)) ? 1 : 0;
        if (pool && !pp) return RSU_EINVAL;
        auto apply_plan = [&](IgFwdParams& q, const Fwd2Plan& pl) {
            q.ncob = pl.ncob;
            q.g = pl.g;
            q.lsw = pl.lsw;
            q.ksplit = pl.ksplit;
            q.cob_group = pl.cob_group;
            q.kslab = pl.ksplit > 1 ? kws : nullptr;
            q.kslab_stride = pl.ksplit > 1 ? (long)igemm_pp_slab_floats(pl.cfg, q) : 0;
        };
        apply_plan(p, pl2);
        if (p.ksplit > 1 && (!pp || (size_t)p.ksplit * (size_t)p.kslab_stride > kws_floats)) return RSU_EINVAL;
#ifdef RSU_DEV_KERNELS   // developer build only: timing ablations and the time-stamping kernels (the default build ignores these variables)
        p.dbg = env_int("RSU_FWD_DBG", 0);
        if (p.dbg & 128) {
            const char* sp = getenv("RSU_STAMP_PTR");
            p.stamps = sp ? (unsigned*)strtoull(sp, nullptr, 0) : nullptr;
        }
#endif
        if (env_int("RSU_PLAN_DEBUG", 0)) {
            const IgFwdCfgInfo ci = igemm_fwd2_cfg_info(pl2.cfg);
            const long tiles = (long)N * pl2.g.nstrips * pl2.g.tiles_per_strip;
            const long workers = pl2.grid_x / (pl2.ncob * pl2.ksplit);
            const long rounds = (tiles + workers - 1) / workers;
            fprintf(stderr, "[plan fwd2] N%d %dx%d Cout%d ntap%d pad%d: cfg%d TN%d TM%d SW%d strips%d tps%d halo%d tiles%ld grid%d rounds%ld ksplit%d cgrp%d pix_util %.3f total_util %.3f\n",
                    N, Ho, Wo, Cout, ntap, pad, pl2.cfg, ci.TN, ci.TM, pl2.g.SW, pl2.g.nstrips, pl2.g.tiles_per_strip, pl2.g.npix_max, tiles,
                    pl2.grid_x, rounds, pl2.ksplit, pl2.cob_group, (double)N * Ho * Wo / ((double)tiles * ci.TM),
                    (double)N * Ho * Wo * Cout / ((double)rounds * pl2.grid_x * ci.TM * ci.TN) * pl2.ksplit);
        }
        if (tune_now && !split) {
            // time every admissible shape of the same channel-block width (1 untimed + 5 timed launches each, fastest counts, device idle; the
            // launches all write the same values, so the output is valid whichever ran last)
            const int tn_model = igemm_fwd2_cfg_info(pl2.cfg).TN;
            struct EventPair {  // destroyed on every exit path
                hipEvent_t a = nullptr, b = nullptr;
                ~EventPair() {
                    if (a) (void)hipEventDestroy(a);
                    if (b) (void)hipEventDestroy(b);
                }
            } ev;
            HIP_CHECK_RET(hipEventCreate(&ev.a));
            HIP_CHECK_RET(hipEventCreate(&ev.b));
            const hipEvent_t e0 = ev.a, e1 = ev.b;
            HIP_CHECK_RET(hipDeviceSynchronize());
            float best_ms = 1e30f, model_ms = 1e30f;
            int best_cfg = pl2.cfg | (pp << 8);
            const int model_code = best_cfg;
            for (int cfg = 0; cfg < IGF2_NCFG; ++cfg) {
                if (igemm_fwd2_cfg_info(cfg).TN != tn_model) continue;
                Fwd2Plan pc;
                if (!plan_fwd2(pc, ncu, N, Ho, Wo, Cout, ntap, kh, kw, dil, stride, gy, ktot, cfg, shared_chip, pool)) continue;
                IgFwdParams pt = p;
                apply_plan(pt, pc);
                for (int vpp = 0; vpp < 2; ++vpp) {  // the kernel generations of the shape
                    if (vpp == 1 && !(pp_ok && !accumulate && pp_supports(cfg, pt))) continue;
                    if (vpp == 0 && pool) continue;   // (only the ping-pong kernel folds the pool)
                    if (!vpp && gen >= 4 && pp_ok && !accumulate && pp_supports(cfg, pt)) continue;
                    float ms_min = 1e30f;
                    for (int rep = 0; rep < 6; ++rep) {
                        HIP_CHECK_RET(hipEventRecord(e0, st));
                        HIP_CHECK_RET(launch_persistent(vpp, pc.cfg, ntap, pt, pc.grid_x, gy, st));
                        HIP_CHECK_RET(hipEventRecord(e1, st));
                        HIP_CHECK_RET(hipEventSynchronize(e1));
                        float ms = 0.f;
                        HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 && ms < ms_min) ms_min = ms;
                    }
                    if (env_int("RSU_PLAN_DEBUG", 0)) fprintf(stderr, "[tune fwd2] cfg%d pp%d %.1f us\n", cfg, vpp, ms_min * 1e3f);
                    const int code = cfg | (vpp << 8);
                    if (code == model_code) model_ms = ms_min;
                    if (ms_min < best_ms) {
                        best_ms = ms_min;
                        best_cfg = code;
                    }
                }
            }
            // hysteresis: a shape replaces the cost model's choice only when it measured at least 3 % faster (one noisy sample must
            // not pin a slow shape for the rest of the run)
            if (best_ms > 0.97f * model_ms) best_cfg = model_code;
            if (env_int("RSU_PLAN_DEBUG", 0)) fprintf(stderr, "[tune fwd2] model cfg%d pp%d -> measured cfg%d pp%d\n", pl2.cfg, (int)pp, best_cfg & 255, best_cfg >> 8);
            std::lock_guard<std::mutex> lk(g_tune_mutex);
            g_tuned[tkey] = best_cfg;
            return RSU_OK;
        }
        HIP_CHECK_RET(launch_persistent(pp, pl2.cfg, ntap, p, pl2.grid_x, gy, st));
        return RSU_OK;
    }
}

// workspace a launch needs before it may cut its reduction into slices (split-K): room for one fp32 copy of the padded output per slice on
// a chip's worth of workgroups -- 256 workgroups x the largest accumulator tile (128 x 256 floats)
extern "C" size_t rsu_conv_splitk_ws_floats(void) { return (size_t)256 * 128 * 256 + 1024; }
extern "C" int rsu_conv2d_fwd_k(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, int N, int Hin,
                                int Win, int Cout, int dil, int relu, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream) {
    if (!srcs || nsrc < 1 || nsrc > 3 || !packed_fwd || !y || Cout % 8 || (dil != 1 && dil != 2)) return RSU_EINVAL;
    const int Ho = Hin - 2 * dil, Wo = Win - 2 * dil;
    if (Ho < 1 || Wo < 2) return RSU_EINVAL;
    return run_fwd(srcs, nsrc, packed_fwd, 0, rup(Cout, 128) / 16, 0, bias, y, nullptr, N, Hin, Win, Ho, Wo, Cout, Cout, 9, 3, dil, 1, 0,
                   Ho, Wo, 1, 1, relu, 0, ncu, (hipStream_t)stream, nullptr, nullptr, kws, kws_floats);
}
extern "C" int rsu_conv2d_fwd(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, int N, int Hin,
                              int Win, int Cout, int dil, int relu, int ncu, rsu_stream_t stream) {
    return rsu_conv2d_fwd_k(srcs, nsrc, packed_fwd, bias, y, N, Hin, Win, Cout, dil, relu, ncu, nullptr, 0, stream);
}

extern "C" int rsu_conv2d_fwd_pool_k(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, void* pooled, void* code,
                                     int N, int Hin, int Win, int Cout, float keep, unsigned key, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream);
extern "C" int rsu_conv2d_fwd_pool(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, void* pooled, void* code,
                                   int N, int Hin, int Win, int Cout, float keep, unsigned key, int ncu, rsu_stream_t stream) {
    return rsu_conv2d_fwd_pool_k(srcs, nsrc, packed_fwd, bias, y, pooled, code, N, Hin, Win, Cout, keep, key, ncu, nullptr, 0, stream);
}
extern "C" int rsu_conv2d_fwd_pool_k(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, void* pooled, void* code,
                                     int N, int Hin, int Win, int Cout, float keep, unsigned key, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream) {
    if (!srcs || nsrc < 1 || nsrc > 3 || !packed_fwd || !y || !pooled || Cout % 8 || !keep_ok(keep)) return RSU_EINVAL;
    const int Ho = Hin - 2, Wo = Win - 2;
    if (Ho < 2 || Wo < 2) return RSU_EINVAL;
    if (code && ((Ho | Wo) & 1)) return RSU_EINVAL;   // (the code bytes describe whole 2x2 windows: refused before anything is launched)
    const bool even = ((Ho | Wo) & 1) == 0;   // the folded epilogue pools whole 2x2 windows; odd sizes take the two launches (floor semantics, no code bytes)
    // the pool folds into the conv's epilogue where a ping-pong tile shape with whole window rows per wave fits the layer and no dropout
    // follows (its mask is a function of the pooled element index: the separate kernel applies it); otherwise: the two launches
    // ... and where the shapes that can fold it (strip width 16 / 32) cost the conv no extra round of tiles: the pool kernel it saves
    // is short (measured: profiles/r03/pool_fusion.txt; RSU_POOL_FUSED=2 folds wherever a shape exists, 0 never)
    const int fuse = env_int("RSU_POOL_FUSED", 1);
    bool worth = fuse >= 2;
    if (fuse == 1) {
        Fwd2Plan a, b;
        const int n = launch_ncu(ncu), kt = 0;
        int ktot = 0;
        for (int i = 0; i < nsrc; ++i) ktot += rup(srcs[i].C, 32) * 9;
        (void)kt;
        Fwd2Plan sp;
        worth = n > 0 && !(kws && plan_split(sp, N, Ho, Wo, Cout, ktot, kws_floats)) &&   // (a layer that splits its reduction does not fold the pool)
                plan_fwd2(a, n, N, Ho, Wo, Cout, 9, 3, 3, 1, 1, 1, ktot, -1, false, false) &&
                plan_fwd2(b, n, N, Ho, Wo, Cout, 9, 3, 3, 1, 1, 1, ktot, -1, false, true) && b.cost <= 1.08 * a.cost;
    }
    if (keep == 1.f && even && worth && env_int("RSU_FWD_GEN", 3) >= 3 && env_int("RSU_FWD2_CFG", -1) < 0) {
        const int rc = run_fwd(srcs, nsrc, packed_fwd, 0, rup(Cout, 128) / 16, 0, bias, y, nullptr, N, Hin, Win, Ho, Wo, Cout, Cout, 9, 3, 1, 1, 0, Ho, Wo,
                               1, 1, 1, 0, ncu, (hipStream_t)stream, pooled, code);
        if (rc != RSU_EINVAL) return rc;
    }
    const int rc = rsu_conv2d_fwd_k(srcs, nsrc, packed_fwd, bias, y, N, Hin, Win, Cout, 1, 1, ncu, kws, kws_floats, stream);
    if (rc != RSU_OK) return rc;
    return rsu_maxpool2x2_fwd_code(y, pooled, code, N, Ho, Wo, Cout, keep, key, stream);
}

extern "C" size_t rsu_packed_first_bytes(int Cout) {
    int seg[1] = {3};
    return rsu_packed_bytes(9, Cout, seg, 1);
}
extern "C" int rsu_pack_conv_first(const float* w, void* packed, int Cout, rsu_stream_t stream) {
    int seg[1] = {3};
    return do_pack(w, packed, 9, Cout, seg, 1, (long)3 * Cout, 1, Cout, 0, (hipStream_t)stream);
}
extern "C" int rsu_conv_first_fwd(const void* in16, const void* packed, const float* b, void* y, int N, int H, int W, int Cout, int dil,
                                  int ncu, rsu_stream_t stream) {
    if (!in16 || !packed || !y || Cout % 8 || H <= 2 * dil || W <= 2 * dil + 1 || (dil != 1 && dil != 2)) return RSU_EINVAL;
    rsu_src_t s;
    s.ptr = in16; s.H = H; s.W = W; s.C = 16; s.oy = 0; s.ox = 0;
    const int Ho = H - 2 * dil, Wo = W - 2 * dil;
    if (env_int("RSU_FIRST_GEN", 1) != 0) {   // (0: the layer as a launch of the generic implicit-GEMM kernels, as in rounds 1-3a)
        const int n = launch_ncu(ncu);
        if (n < 0) return RSU_EINVAL;
        if ((long)N * Ho * Wo * Cout * 2 >= 0x7ffffff0L || (long)N * H * W * 32 >= 0x7ffffff0L) return RSU_E2BIG;
        HIP_CHECK_RET(conv_first_fwd_launch(in16, packed, rup(Cout, 128) / 16, b, y, N, H, W, Cout, dil, 1, n, (hipStream_t)stream));
        return RSU_OK;
    }
    return run_fwd(&s, 1, packed, 0, rup(Cout, 128) / 16, 0, b, y, nullptr, N, H, W, Ho, Wo, Cout, Cout, 9, 3, dil, 1, 0, Ho, Wo, 1, 1, 1, 0,
                   ncu, (hipStream_t)stream);
}

extern "C" int rsu_color_conv_first_fwd(const float* x, const float* w0, const float* b0, const void* packed, const float* b, void* y, int N, int H,
                                        int W, int Cout, int dil, int ncu, rsu_stream_t stream) {
    if (!x || !w0 || !b0 || !packed || !y || Cout % 8 || H <= 2 * dil || W <= 2 * dil + 1 || (dil != 1 && dil != 2)) return RSU_EINVAL;
    const int Ho = H - 2 * dil, Wo = W - 2 * dil;
    const int n = launch_ncu(ncu);
    if (n < 0) return RSU_EINVAL;
    if ((long)N * Ho * Wo * Cout * 2 >= 0x7ffffff0L || (long)N * H * W * 12 >= 0x7ffffff0L) return RSU_E2BIG;
    HIP_CHECK_RET(conv_first_fwd_launch(nullptr, packed, rup(Cout, 128) / 16, b, y, N, H, W, Cout, dil, 1, n, (hipStream_t)stream, x, w0, b0));
    return RSU_OK;
}

extern "C" int rsu_conv2d_bwd_data_k(const void* dz, const void* packed_bwd, void* dx, const void* relu_src, int accumulate, int N, int H,
                                     int W, int Cin_total, int ci_off, int ci_cnt, int Cout, int dil, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream);
extern "C" int rsu_conv2d_bwd_data(const void* dz, const void* packed_bwd, void* dx, const void* relu_src, int accumulate, int N, int H,
                                   int W, int Cin_total, int ci_off, int ci_cnt, int Cout, int dil, int ncu, rsu_stream_t stream) {
    return rsu_conv2d_bwd_data_k(dz, packed_bwd, dx, relu_src, accumulate, N, H, W, Cin_total, ci_off, ci_cnt, Cout, dil, ncu, nullptr, 0, stream);
}
extern "C" int rsu_conv2d_bwd_data_k(const void* dz, const void* packed_bwd, void* dx, const void* relu_src, int accumulate, int N, int H,
                                     int W, int Cin_total, int ci_off, int ci_cnt, int Cout, int dil, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream) {
    if (!dz || !packed_bwd || !dx || Cout % 8 || ci_cnt % 8 || ci_off % 32 || ci_off + ci_cnt > Cin_total || (dil != 1 && dil != 2))
        return RSU_EINVAL;
    const int Hd = H - 2 * dil, Wd = W - 2 * dil;  // dz size
    if (Hd < 1 || Wd < 1 || W < 2) return RSU_EINVAL;
    rsu_src_t s;
    s.ptr = dz; s.H = Hd; s.W = Wd; s.C = Cout; s.oy = 0; s.ox = 0;
    return run_fwd(&s, 1, packed_bwd, 0, rup(Cin_total, 128) / 16, ci_off / 16, nullptr, dx, relu_src, N, Hd, Wd, H, W, ci_cnt, ci_cnt, 9,
                   3, dil, 1, 2 * dil, H, W, 1, 1, 0, accumulate, ncu, (hipStream_t)stream, nullptr, nullptr, kws, kws_floats);
}

// igemm_ct launch (transposed conv forward / backward-data as a ping-pong GEMM): one workgroup per budgeted CU, column blocks fastest
static unsigned magic_floor32(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }
static int run_ct(int mode, const void* a, int Ca, int N, int H, int W, const void* wp, long phase_stride, int ntiles_w, const float* bias,
                  void* out, int outC, int Cn, const void* mask_src, int ncu_arg, hipStream_t st) {
    const int ncu = launch_ncu(ncu_arg);
    if (ncu < 0) return RSU_EINVAL;
    IgCtParams p;
    memset(&p, 0, sizeof(p));
    p.a = (const bf16_t*)a;
    p.Ca = Ca;
    p.nchunk = cdiv(Ca, 32);
    p.N = N; p.H = H; p.W = W;
    p.magic_hw = magic_floor32(H * W);
    p.magic_w = magic_floor32(W);
    p.wp = (const bf16_t*)wp;
    p.wp_phase_stride = phase_stride;
    p.ntiles_w = ntiles_w;
    p.bias = bias;
    p.out = (bf16_t*)out;
    p.outC = outC;
    p.Cn = Cn;
    p.mask_src = (const bf16_t*)mask_src;
    p.nnb = cdiv((mode == 0 ? 2 : 1) * Cn, 128);
    p.ncob = (mode == 0 ? 2 : 1) * p.nnb;
    const int ntile_m = cdiv(N * H * W, 256);
    int cols = ncu / p.ncob;   // workgroups per column block
    if (cols < 1) cols = 1;
    if (cols > ntile_m) cols = ntile_m;
    HIP_CHECK_RET(igemm_ct_launch(mode, p, cols * p.ncob, st));
    return RSU_OK;
}

extern "C" int rsu_convT2x2_fwd(const void* x, const void* packed_fwd, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                                int ncu, rsu_stream_t stream) {
    if (!x || !packed_fwd || !y || Cin % 8 || Cout % 8 || W < 2) return RSU_EINVAL;
    rsu_src_t s;
    s.ptr = x; s.H = H; s.W = W; s.C = Cin; s.oy = 0; s.ox = 0;
    int seg[1] = {Cin};
    const long per = (long)(rsu_packed_bytes(1, Cout, seg, 1) / 2);
    // RSU_CT_GEN=1: the 1-tap launches of igemm_fwd2 (one per output phase); default: the ping-pong GEMM of igemm_ct.hip
    if (env_int("RSU_CT_GEN", 2) >= 2 && igemm_ct_supports(0, N, H, W, Cin, Cout) && (long)N * 4 * H * W * Cout * 2 < 0x7ffffff0L)
        return run_ct(0, x, Cin, N, H, W, packed_fwd, per, rup(Cout, 128) / 16, bias, y, Cout, Cout, nullptr, ncu, (hipStream_t)stream);
    return run_fwd(&s, 1, packed_fwd, per, rup(Cout, 128) / 16, 0, bias, y, nullptr, N, H, W, H, W, Cout, Cout, 1, 1, 1, 1, 0, 2 * H, 2 * W,
                   2, 4, 0, 0, ncu, (hipStream_t)stream);
}

extern "C" int rsu_convT2x2_bwd_data(const void* dy, const void* packed_bwd, void* dx, const void* relu_src, float out_scale, int N, int H,
                                     int W, int Cin, int Cout, int ncu, rsu_stream_t stream) {
    if (!dy || !packed_bwd || !dx || Cin % 8 || Cout % 8 || W < 2 || !(out_scale > 0.f)) return RSU_EINVAL;
    rsu_src_t s;
    s.ptr = dy; s.H = 2 * H; s.W = 2 * W; s.C = Cout; s.oy = 0; s.ox = 0;
    const bool ct = env_int("RSU_CT_GEN", 2) >= 2 && igemm_ct_supports(1, N, H, W, Cout, Cin) && (long)N * 4 * H * W * Cout * 2 < 0x7ffffff0L;
    const int rc = ct ? run_ct(1, dy, Cout, N, H, W, packed_bwd, 0, rup(Cin, 128) / 16, nullptr, dx, Cin, Cin, relu_src, ncu, (hipStream_t)stream)
                      : run_fwd(&s, 1, packed_bwd, 0, rup(Cin, 128) / 16, 0, nullptr, dx, relu_src, N, 2 * H, 2 * W, H, W, Cin, Cin, 4, 2, 1, 2, 0,
                                H, W, 1, 1, 0, 0, ncu, (hipStream_t)stream);
    if (rc != RSU_OK || out_scale == 1.f) return rc;
    // 1/keep of a dropout in front of the transposed conv: a separate pass over the (small) gradient tensor, training with
    // dropout only -- the MFMA kernel's epilogue is left alone (see DESIGN.md section 4 on what scaling there cost)
    HIP_CHECK_RET(ew_scale_bf16(dx, (long)N * H * W * Cin, out_scale, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// igemm_wgrad family
// ---------------------------------------------------------------------------------------------
// the 64x64 shape is widened to 128 F channels per workgroup whenever F has that many (RSU_WG_CFG=0 keeps 64x64, A/B runs)
static int wgrad_pick_cfg(int cfg, int Cf) {
    if (cfg == IGW_CFG_64x64 && Cf >= 128 && env_int("RSU_WG_CFG", IGW_CFG_128x64) == IGW_CFG_128x64) return IGW_CFG_128x64;
    return cfg;
}
// grid.z splits of the pixel reduction: one workgroup per CU in total
static int wgrad_max_split(int cfg, int Cf, int Cs, int ncu = 256) {
    int want = ncu / (cdiv(Cf, igemm_wgrad_cfb(cfg)) * cdiv(Cs, igemm_wgrad_csb(cfg)));
    return want < 1 ? 1 : want;
}
// slabs the workspace must hold (either shape the launch may pick)
static size_t wgrad_max_slabs(int cfg, int Cf, int Cs) {
    const int a = wgrad_max_split(cfg, Cf, Cs), b = wgrad_max_split(wgrad_pick_cfg(cfg, Cf), Cf, Cs);
    return (size_t)(a > b ? a : b);
}
struct WgPlan { int cfg; TileGeo g; int gx, gy, nsplit, ntiles, lsw, nbuf; };
static bool plan_wgrad(WgPlan& pl, int ncu, int cfg, int N, int Hf, int Wf, int Cf, int Cs, int ntap, int kh, int kw, int dil, int stride) {
    pl.cfg = cfg;
    const int csb = igemm_wgrad_csb(cfg), cfbk = igemm_wgrad_cfb(cfg);
    const int tmk = igemm_wgrad_tmk(cfg);
    // three staging buffers (loads two tiles ahead) when a halo tile small enough exists, else two
    bool ok = false;
    for (int nbuf = 3; nbuf >= 2 && !ok; --nbuf) {
        const long fixed = (long)nbuf * tmk * 2 * cfbk;
        if (fixed >= 150 * 1024) continue;
        int cap = (int)((160 * 1024 - fixed) / (nbuf * csb * 2));
        const int ppw = 64 / (csb / 8) * 8;  // S pixels covered by one piece per wave
        cap = cap / ppw * ppw;               // the kernel rounds the S slot up to whole pieces per wave
        const int cap5 = 5 * ppw;            // ... and knows wait counts for at most 5 of them
        if (nbuf == 3 && cap > cap5) cap = cap5;
        if (cap < ppw) continue;
        pl.nbuf = nbuf;
        ok = plan_geo_aligned(pl.g, pl.lsw, Hf, Wf, tmk, kh, kw, dil, stride, cap);
    }
    if (!ok) return false;
    pl.gx = cdiv(Cf, cfbk);
    pl.gy = cdiv(Cs, csb);
    pl.ntiles = N * pl.g.nstrips * pl.g.tiles_per_strip;
    const int want = wgrad_max_split(cfg, Cf, Cs, ncu);  // workgroups along z; each writes one slab (workspace sized for 256)
    pl.nsplit = want < pl.ntiles ? want : pl.ntiles;
    (void)ntap;
    return true;
}

// One weight-gradient launch, prepared: kernel shape, geometry, every parameter but the pixel split and the output pointers.
struct WgPrep {
    int cfg, ntap, gx, gy, ntiles, extra;   // extra: floats behind the taps of a slab (bias sums)
    long main_elems;
    IgWgradParams p;
    float *out, *db, *dbs;
    int cs_cnt;
};
static int prep_wgrad(WgPrep& w, int cfg, const void* F, int Hf, int Wf, int Cf, const rsu_src_t* S, float* out, int CsOut, int CfOut,
                      int cs_off, int N, int ntap, int kw, int dil, int stride, int ncu, float* db, float* dbs) {
    // 32-bit byte offsets inside the kernel: both operand tensors must stay below 2 GiB (beyond that the offsets wrapped
    // silently and only the weight gradients came out wrong)
    if ((long)N * Hf * Wf * Cf * 2 >= 0x7ffffff0L || (long)N * S->H * S->W * S->C * 2 >= 0x7ffffff0L) return RSU_E2BIG;
    WgPlan pl;
    const int wide = wgrad_pick_cfg(cfg, Cf);
    if (wide != cfg && plan_wgrad(pl, ncu, wide, N, Hf, Wf, Cf, S->C, ntap, ntap / kw, kw, dil, stride)) cfg = wide;  // else: halo too big for LDS
    else if (!plan_wgrad(pl, ncu, cfg, N, Hf, Wf, Cf, S->C, ntap, ntap / kw, kw, dil, stride)) return RSU_EINVAL;
    if (db && dbs) return RSU_EINVAL;
    if (dbs && (CsOut % 4 || cs_off != 0 || S->C != CsOut)) return RSU_EINVAL;
    IgWgradParams& p = w.p;
    memset(&p, 0, sizeof(p));
    p.F = (const bf16_t*)F;
    p.Hf = Hf; p.Wf = Wf; p.Cf = Cf;
    p.S.ptr = (const bf16_t*)S->ptr;
    p.S.H = S->H; p.S.W = S->W; p.S.C = S->C; p.S.oy = S->oy; p.S.ox = S->ox;
    p.CsOut = CsOut; p.CfOut = CfOut; p.cs_off = cs_off;
    p.zero_page = zero_page();
    if (!p.zero_page) return RSU_EHIP;
    p.N = N; p.dil = dil; p.stride = stride;
    p.nsplit = pl.nsplit;   // the split of a launch that has the chip to itself (finish_wgrad may set another)
    p.ntiles_total = pl.ntiles;
    p.lsw = pl.lsw;
    p.nbuf = pl.nbuf;
    p.nsw = igemm_wgrad_nsw(cfg, pl.g.npix_max);
#ifdef RSU_DEV_KERNELS
    p.dbg = env_int("RSU_WG_DBG", 0);
#endif
    p.g = pl.g;
    w.cfg = cfg; w.ntap = ntap; w.gx = pl.gx; w.gy = pl.gy; w.ntiles = pl.ntiles;
    w.main_elems = (long)ntap * CsOut * CfOut;
    // extra items behind the taps of each slab: the F column sums (db, CfOut floats) or the S column sums (dbs, CsOut floats)
    w.extra = db ? CfOut : (dbs ? rup(CsOut, 4) : 0);
    w.out = out; w.db = db; w.dbs = dbs; w.cs_cnt = S->C;
    return RSU_OK;
}
// the pixel split and where the partial results go: a single split needs no slab -- its workgroups write the gradient (and the bias
// sums) in place; otherwise the bias sums are one more row behind the taps of each slab, so a single reduce finishes both
static void finish_wgrad(WgPrep& w, int nsplit, float* ws) {
    IgWgradParams& p = w.p;
    p.nsplit = nsplit;
    p.slab = nsplit == 1 ? w.out : ws;
    p.slab_stride = w.main_elems + w.extra;
    p.bslab = w.db ? (nsplit == 1 ? w.db : ws + w.main_elems) : nullptr;
    p.sbslab = w.dbs ? (nsplit == 1 ? w.dbs : ws + w.main_elems) : nullptr;
}
struct WgUpdate { const UpJob* job; int seg, keep_grad; float lr, mu, gscale; };
static int run_wgrad(int cfg, const void* F, int Hf, int Wf, int Cf, const rsu_src_t* S, float* out, float* ws, int CsOut, int CfOut,
                     int cs_off, int N, int ntap, int kw, int dil, int stride, int ncu_arg, hipStream_t st, float* db = nullptr, float* dbs = nullptr,
                     const WgUpdate* upd = nullptr) {
    const int ncu = launch_ncu(ncu_arg);
    if (ncu < 0) return RSU_EINVAL;
    WgPrep w;
    const int rc = prep_wgrad(w, cfg, F, Hf, Wf, Cf, S, out, CsOut, CfOut, cs_off, N, ntap, kw, dil, stride, ncu, db, dbs);
    if (rc != RSU_OK) return rc;
    const int nslab = w.p.nsplit;
    finish_wgrad(w, nslab, ws);
    const IgWgradParams& p = w.p;
    // RSU_WG_GEN=1: igemm_wgrad everywhere; default: the ping-pong kernel where it is built (same slabs, same bits)
    if (env_int("RSU_WG_GEN", 2) >= 2 && igemm_wgpp_supports(w.cfg, ntap, p))
        HIP_CHECK_RET(igemm_wgpp_launch(p, w.gx, w.gy, nslab, st));
    else if (env_int("RSU_WG_GEN", 2) >= 2 && env_int("RSU_WG64", 1) && igemm_wgp64_supports(w.cfg, ntap, p))
        HIP_CHECK_RET(igemm_wgp64_launch(p, w.gx, w.gy, nslab, st));
    else
        HIP_CHECK_RET(igemm_wgrad_launch(w.cfg, ntap, p, w.gx, w.gy, nslab, st));
    if (upd) {
        // the reduce launch IS the Momentum step + re-pack of the rows this source owns (k_update_pack_seg): nslab == 1 reads the gradient
        // the kernel above wrote in place
        // -- unless the launch has many slabs of a small kernel (>= 16: the 64- .. 256-channel layers): the update pass has one workgroup per
        // 32 x 128 weights, far too few to stream 16 .. 128 slabs each (measured: c2 945 -> 750 patches/s); those keep the wide reduce launch and the
        // update reads the finished gradient, L2-hot
        const int fuse_max = env_int("RSU_FUSED_MAX_SPLIT", 15);
        if (nslab > fuse_max) {
            HIP_CHECK_RET(ew_reduce_slabs(ws, out, db ? db : dbs, w.extra / 4, nslab, p.slab_stride, ntap, CsOut, cs_off, S->C, CfOut, st));
            HIP_CHECK_RET(ew_update_pack_seg(*upd->job, upd->seg, out, 0, 1, nullptr, nullptr, 0, upd->lr, upd->mu, upd->gscale, st));
        } else {
            HIP_CHECK_RET(ew_update_pack_seg(*upd->job, upd->seg, nslab > 1 ? ws : out, nslab > 1 ? p.slab_stride : 0, nslab, upd->keep_grad ? out : nullptr,
                                             db ? db : dbs, w.extra / 4, upd->lr, upd->mu, upd->gscale, st));
        }
    } else if (nslab > 1) {
        HIP_CHECK_RET(ew_reduce_slabs(ws, out, db ? db : dbs, w.extra / 4, nslab, p.slab_stride, ntap, CsOut, cs_off, S->C, CfOut, st));
    }
    return RSU_OK;
}

extern "C" size_t rsu_conv2d_bwd_weight_ws_floats(int Cin_total, int src_C, int Cout) {
    return wgrad_max_slabs(IGW_CFG_64x64, Cout, src_C) * (9 * (size_t)Cin_total * Cout + Cout);
}
extern "C" int rsu_conv2d_bwd_weight(const rsu_src_t* src, const void* dz, float* dw, float* db, float* ws, int N, int Ho, int Wo,
                                     int Cin_total, int ci_off, int Cout, int dil, int ncu, rsu_stream_t stream) {
    if (!src || !src->ptr || !dz || !dw || !ws || src->C % 8 || Cout % 8 || ci_off + src->C > Cin_total || (dil != 1 && dil != 2))
        return RSU_EINVAL;
    if (src->oy < 0 || src->ox < 0 || src->oy + Ho + 2 * dil > src->H || src->ox + Wo + 2 * dil > src->W || Wo < 2) return RSU_EINVAL;
    // F = dz (cf = co), S = layer input (cs = ci): slab[tap][ci][co] = HWIO
    return run_wgrad(IGW_CFG_64x64, dz, Ho, Wo, Cout, src, dw, ws, Cin_total, Cout, ci_off, N, 9, 3, dil, 1, ncu, (hipStream_t)stream, db);
}

extern "C" int rsu_conv2d_bwd_weight_update(const rsu_src_t* src, const void* dz, float* dw, float* db, float* ws, int N, int Ho, int Wo,
                                            int Cin_total, int ci_off, int Cout, int dil, int ncu, const void* update_entry, int seg, float lr,
                                            float mu, float gscale, int keep_grad, rsu_stream_t stream) {
    if (!src || !src->ptr || !dz || !dw || !ws || src->C % 8 || Cout % 8 || ci_off + src->C > Cin_total || (dil != 1 && dil != 2))
        return RSU_EINVAL;
    if (src->oy < 0 || src->ox < 0 || src->oy + Ho + 2 * dil > src->H || src->ox + Wo + 2 * dil > src->W || Wo < 2) return RSU_EINVAL;
    const UpJob* J = (const UpJob*)update_entry;
    // the entry must describe THIS kernel (rsu_update_table_add, kind RSU_PACK_CONV_FWD, g = dw) and `seg` the source at hand
    if (!J || J->kind != 1 || J->ntap != 9 || J->R1 != Cin_total || J->R2 != Cout || J->g != dw || seg < 0 || seg >= J->nseg ||
        J->seg_r0[seg] != ci_off || J->seg_c[seg] != src->C)
        return RSU_EINVAL;
    WgUpdate u{J, seg, keep_grad, lr, mu, gscale};
    return run_wgrad(IGW_CFG_64x64, dz, Ho, Wo, Cout, src, dw, ws, Cin_total, Cout, ci_off, N, 9, 3, dil, 1, ncu, (hipStream_t)stream, db, nullptr, &u);
}

extern "C" size_t rsu_convT2x2_bwd_weight_ws_floats(int Cin, int Cout) {
    return wgrad_max_slabs(IGW_CFG_64x64, Cin, Cout) * (4 * (size_t)Cout * Cin + Cout);
}
extern "C" int rsu_convT2x2_bwd_weight(const void* x, const void* dy, float* dK, float* db, float* ws, int N, int H, int W, int Cin,
                                       int Cout, int ncu, rsu_stream_t stream) {
    if (!x || !dy || !dK || !ws || Cin % 8 || Cout % 8 || W < 2) return RSU_EINVAL;
    // RSU_WGT_GEN=1: the generic igemm_wgrad launch (4 taps, stride 2) of rounds 1-3a; 3 (default since round 6): the ping-pong kernel of
    // igemm_wgt.hip at every CU budget; 2 (rounds 3-5): the ping-pong kernel only where the launch has (most of) the chip to itself. In round 3
    // the faster kernel beside a backward-data kernel made the STEP slower (945 -> 938 patches/s, profiles/r03/lib_ab_wgt.txt); with round 4's
    // backward-data kernels on the other half of the chip the same A/B reads 933 -> 955 (three alternations, profiles/r06/abenv_wgt_gen.txt): the
    // four launches leave the weight-gradient stream -- the longer one, the main stream waits ~60 us for it at the end of the pass -- 0.14 ms shorter.
    const int wgt_gen = env_int("RSU_WGT_GEN", 3);
    if (wgt_gen >= 2 && igemm_wgt_supports(N, H, W, Cin, Cout) && (wgt_gen >= 3 || launch_ncu(ncu) >= 192)) {
        const int n = launch_ncu(ncu);
        if (n < 0) return RSU_EINVAL;
        const int ntiles = igemm_wgt_tiles(N, H, W);
        int nsplit = n / igemm_wgt_blocks(Cin, Cout);   // one workgroup per budgeted CU (the workspace holds the splits of a whole chip)
        nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
        const long main_elems = 4l * Cout * Cin, extra = db ? rup(Cout, 4) : 0, stride = main_elems + extra;
        HIP_CHECK_RET(igemm_wgt_launch(x, dy, nsplit == 1 ? dK : ws, db ? (nsplit == 1 ? db : ws + main_elems) : nullptr, stride, N, H, W, Cin, Cout,
                                       nsplit, (hipStream_t)stream));
        if (nsplit > 1) HIP_CHECK_RET(ew_reduce_slabs(ws, dK, db, (int)(extra / 4), nsplit, stride, 4, Cout, 0, Cout, Cin, (hipStream_t)stream));
        return RSU_OK;
    }
    rsu_src_t s;
    s.ptr = dy; s.H = 2 * H; s.W = 2 * W; s.C = Cout; s.oy = 0; s.ox = 0;
    // F = x (cf = ci), S = dy (cs = co), stride 2: slab[tap(a,b)][co][ci] = K layout
    return run_wgrad(IGW_CFG_64x64, x, H, W, Cin, &s, dK, ws, Cout, Cin, 0, N, 4, 2, 1, 2, ncu, (hipStream_t)stream, nullptr, db);
}

extern "C" size_t rsu_conv_first_bwd_ws_floats(int Cout) {
    return (wgrad_max_slabs(IGW_CFG_64x16, Cout, 16) + 1) * (9 * 16 * (size_t)Cout + Cout);
}
extern "C" int rsu_conv_first_bwd_weight(const void* in16, const void* dz, float* dw1, float* gxc, float* db, float* ws, int N, int H, int W,
                                         int Cout, int dil, int ncu, rsu_stream_t stream) {
    if (!in16 || !dz || !dw1 || !ws || Cout % 8 || (dil != 1 && dil != 2) || W - 2 * dil < 2) return RSU_EINVAL;
    rsu_src_t s;
    s.ptr = in16; s.H = H; s.W = W; s.C = 16; s.oy = 0; s.ox = 0;
    float* tmp = ws;                          // [9][16][Cout]
    float* slabs = ws + (size_t)9 * 16 * Cout;
    // RSU_WG1_GEN=1: the generic igemm_wgrad launch (64x16 shape) of rounds 1-3a; default: the ping-pong kernel of igemm_wg1.hip
    if (env_int("RSU_WG1_GEN", 2) >= 2 && igemm_wg1_supports(N, H, W, Cout, dil)) {
        const int n = launch_ncu(ncu);
        if (n < 0) return RSU_EINVAL;
        const int ntiles = igemm_wg1_tiles(N, H - 2 * dil, W - 2 * dil);
        int nsplit = n / igemm_wg1_blocks(Cout);
        nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
        const long main_elems = 9l * 16 * Cout, extra = db ? Cout : 0, stride = main_elems + extra;
        HIP_CHECK_RET(igemm_wg1_launch(in16, dz, nsplit == 1 ? tmp : slabs, db ? (nsplit == 1 ? db : slabs + main_elems) : nullptr, stride, N, H, W, Cout,
                                       dil, nsplit, (hipStream_t)stream));
        if (nsplit > 1) HIP_CHECK_RET(ew_reduce_slabs(slabs, tmp, db, (int)(extra / 4), nsplit, stride, 9, 16, 0, 16, Cout, (hipStream_t)stream));
        HIP_CHECK_RET(ew_scatter_first_grads(tmp, dw1, gxc, Cout, (hipStream_t)stream));
        return RSU_OK;
    }
    int rc = run_wgrad(IGW_CFG_64x16, dz, H - 2 * dil, W - 2 * dil, Cout, &s, tmp, slabs, 16, Cout, 0, N, 9, 3, dil, 1, ncu, (hipStream_t)stream, db);
    if (rc) return rc;
    HIP_CHECK_RET(ew_scatter_first_grads(tmp, dw1, gxc, Cout, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// grouped weight gradients (igemm_wg_group_kernel): plan once, run every step
// ---------------------------------------------------------------------------------------------
struct WgGroupTable {
    // device-read part (the caller copies the table to the device once; the kernels read it through the constant address space)
    IgWgGroupParams g;
    ReduceJob red[IGW_GROUP_MAX];
    // host-only part
    unsigned magic;
    int nred, red_blocks, nwg_total, nsingle, N, ncu;
    size_t lds_bytes;
    rsu_wgrad_job_t single[IGW_GROUP_MAX];   // jobs no grouped kernel body exists for: launched one by one behind the group
    float* ws;
    double plan_makespan, plan_ideal;        // (diagnostics: tile-steps of the simulated dispatch / of a perfect balance)
};
#define WG_TABLE_MAGIC 0x57474754u
extern "C" size_t rsu_wgrad_group_table_bytes(void) { return sizeof(WgGroupTable); }
// the planner only makes plans whose slabs fit this many floats: up to 512 slabs (two per CU) of a 128 x 64 x 9 block plus a bias row,
// times two (the slab of a concat source spans the rows of all sources of its kernel)
extern "C" size_t rsu_wgrad_group_ws_floats(void) { return (size_t)2 * 512 * (9 * 128 * 64 + 1024) + 4096; }
static int prep_group_job(WgPrep& w, const rsu_wgrad_job_t& j, int N, int ncu) {
    if (j.kind == RSU_WGRAD_CONV3X3) {
        const rsu_src_t* src = &j.src;
        if (!src->ptr || !j.dz || !j.dw || src->C % 8 || j.Cout % 8 || j.ci_off + src->C > j.Cin_total || (j.dil != 1 && j.dil != 2)) return RSU_EINVAL;
        if (src->oy < 0 || src->ox < 0 || src->oy + j.Ho + 2 * j.dil > src->H || src->ox + j.Wo + 2 * j.dil > src->W || j.Wo < 2) return RSU_EINVAL;
        return prep_wgrad(w, IGW_CFG_64x64, j.dz, j.Ho, j.Wo, j.Cout, src, j.dw, j.Cin_total, j.Cout, j.ci_off, N, 9, 3, j.dil, 1, ncu, j.db, nullptr);
    }
    if (j.kind == RSU_WGRAD_CONVT2X2) {   // src = x [N][H][W][Cin], dz = dy [N][2H][2W][Cout]
        const int H = j.src.H, W = j.src.W, Cin = j.src.C, Cout = j.Cout;
        if (!j.src.ptr || !j.dz || !j.dw || Cin % 8 || Cout % 8 || W < 2) return RSU_EINVAL;
        rsu_src_t s;
        s.ptr = j.dz; s.H = 2 * H; s.W = 2 * W; s.C = Cout; s.oy = 0; s.ox = 0;
        return prep_wgrad(w, IGW_CFG_64x64, j.src.ptr, H, W, Cin, &s, j.dw, Cout, Cin, 0, N, 4, 2, 1, 2, ncu, nullptr, j.db);
    }
    return RSU_EINVAL;
}
// relative time of one pixel tile (128 pixels) of one unit, by kernel family (the 128 x 64 x 9 ping-pong kernel = 1)
static double wg_family_cost(int family) {
    if (family >= IGW_FAM_WGPP3 && family <= IGW_FAM_WGPP6) return 1.0;
    if (family == IGW_FAM_WGP64_4 || family == IGW_FAM_WGP64_5) return 0.58;
    switch (family - IGW_FAM_GENERIC) {
        case 2 * IGW_CFG_64x64 + 1: return 0.45;    // 64 x 64 x 4 taps, generic kernel
        case 2 * IGW_CFG_64x16: return 0.30;
        case 2 * IGW_CFG_128x64: return 1.25;       // dilated 3x3, generic kernel
        case 2 * IGW_CFG_128x64 + 1: return 0.62;   // 128 x 64 x 4 taps
    }
    return 1.0;
}
extern "C" int rsu_wgrad_group_plan(const rsu_wgrad_job_t* jobs, int njobs, float* ws, int N, int ncu_arg, void* host_table) {
    if (!jobs || njobs < 1 || njobs > IGW_GROUP_MAX || !ws || !host_table || N < 1) return RSU_EINVAL;
    const int ncu = launch_ncu(ncu_arg);
    if (ncu < 0) return RSU_EINVAL;
    WgGroupTable& T = *(WgGroupTable*)host_table;
    memset(&T, 0, sizeof(T));
    T.magic = WG_TABLE_MAGIC; T.N = N; T.ncu = ncu; T.ws = ws;
    struct Item { WgPrep w; int family, blocks, nsplit, src; double cost, unit; };
    std::array<Item, IGW_GROUP_MAX> it;
    int n = 0;
    double W = 0.0;
    for (int i = 0; i < njobs; ++i) {
        Item& a = it[n];
        const int rc = prep_group_job(a.w, jobs[i], N, ncu);
        if (rc != RSU_OK) return rc;
        a.family = env_int("RSU_WG_GEN", 2) >= 2 ? igemm_wg_group_family(a.w.cfg, a.w.ntap, a.w.p) : -1;
        if (a.family >= 0 && igemm_wg_group_lds_bytes(a.family, a.w.p) > (size_t)IGW_GROUP_LDS_BYTES - 256) a.family = -1;
        if (a.family < 0) {   // no grouped body for this shape: a launch of its own behind the group
            T.single[T.nsingle++] = jobs[i];
            continue;
        }
        a.blocks = a.w.gx * a.w.gy;
        a.src = i;
        a.cost = wg_family_cost(a.family);
        W += (double)a.blocks * a.w.ntiles * a.cost;
        ++n;
    }
    if (n > 0) {
        // Units: job i is cut into blocks_i x nsplit_i units of ~ntiles_i / nsplit_i pixel tiles. The kernel runs ONE unit per workgroup and
        // the dispatcher deals workgroups to CUs as they fall free, so the plan is a list schedule: long units first. Splitting costs a
        // slab per unit (written and read back), so the shallow layers get units of about W / ncu / k tile-steps and the deep layers
        // (many channel blocks, few pixels) none. k is chosen by simulating the dispatch for a few values.
        const double slab_cost = 0.55;   // tile-steps one slab costs its launch (write + its share of the reduce), measured order
        double best = 1e300;
        int best_split[IGW_GROUP_MAX] = {0};
        const double ks[5] = {1.0, 1.5, 2.0, 3.0, 4.0};
        auto search = [&]() {
            best = 1e300;
            for (int ki = 0; ki < 5; ++ki) {
                const double target = W / ncu / ks[ki];
                int split[IGW_GROUP_MAX];
                std::vector<double> units;
                double ws_need = 0.0;
                for (int i = 0; i < n; ++i) {
                    double s = (double)it[i].w.ntiles * it[i].cost / (target > 0 ? target : 1.0);
                    int ns = s < 1.4 ? 1 : (int)(s + 0.5);
                    if (ns > it[i].w.ntiles) ns = it[i].w.ntiles;
                    if (ns < 1) ns = 1;
                    split[i] = ns;
                    const double unit = (double)cdiv(it[i].w.ntiles, ns) * it[i].cost + (ns > 1 ? slab_cost : 0.0);
                    for (int u = 0; u < it[i].blocks * ns; ++u) units.push_back(unit);
                    if (ns > 1) {
                        ws_need += (double)ns * (double)(it[i].w.main_elems + it[i].w.extra);
                    }
                }
                if (ws_need > (double)rsu_wgrad_group_ws_floats()) continue;
                if (units.size() > IGW_UNITS_MAX) continue;
                std::sort(units.begin(), units.end(), [](double a, double b) { return a > b; });
                std::vector<double> cu((size_t)ncu, 0.0);   // greedy dispatch: the next unit goes to the CU that falls free first
                for (double u : units) {
                    size_t m = 0;
                    for (size_t c = 1; c < cu.size(); ++c)
                        if (cu[c] < cu[m]) m = c;
                    cu[m] += u;
                }
                double mk = 0.0;
                for (double c : cu) mk = c > mk ? c : mk;
                if (mk < best) {
                    best = mk;
                    for (int i = 0; i < n; ++i) best_split[i] = split[i];
                }
            }
            return best < 1e300;
        };
        // a group too wide for the unit table or the slab workspace (a deeper or wider net than the benchmarked ones) sheds its largest
        // job into the per-layer launches behind the group until the rest fits: the plan never fails for size alone
        while (n > 0 && !search()) {
            int big = 0;
            for (int i = 1; i < n; ++i)
                if ((double)it[i].blocks * it[i].w.ntiles * it[i].cost > (double)it[big].blocks * it[big].w.ntiles * it[big].cost) big = i;
            W -= (double)it[big].blocks * it[big].w.ntiles * it[big].cost;
            T.single[T.nsingle++] = jobs[it[big].src];
            for (int i = big; i + 1 < n; ++i) it[i] = it[i + 1];
            --n;
        }
        if (n == 0) return RSU_OK;   // (everything went to per-layer launches: rsu_wgrad_group_run issues them)
        T.plan_makespan = best;
        T.plan_ideal = W / ncu;
        for (int i = 0; i < n; ++i) {
            it[i].nsplit = best_split[i];
            it[i].unit = (double)cdiv(it[i].w.ntiles, it[i].nsplit) * it[i].cost;
        }
        // longest units first (stable: ties keep the caller's order)
        std::array<int, IGW_GROUP_MAX> order;
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.begin() + n, [&](int a, int b) { return it[a].unit > it[b].unit; });
        float* wsp = ws;
        int rb = 0;
        struct U { double len; unsigned code; };
        std::vector<U> ulist;
        for (int k = 0; k < n; ++k) {
            Item& a = it[order[k]];
            finish_wgrad(a.w, a.nsplit, wsp);
            IgWgJob& J = T.g.job[k];
            J.p = a.w.p;
            J.gx = a.w.gx; J.gy = a.w.gy; J.gz = a.nsplit;
            J.family = a.family;
            for (int u = 0; u < a.blocks * a.nsplit; ++u) ulist.push_back(U{a.unit + (a.nsplit > 1 ? slab_cost : 0.0), ((unsigned)k << 24) | (unsigned)u});
            const size_t lds = igemm_wg_group_lds_bytes(a.family, a.w.p);
            if (lds > T.lds_bytes) T.lds_bytes = lds;
            if (a.nsplit > 1) {
                ReduceJob& R = T.red[T.nred++];
                R.slab = wsp; R.out = a.w.out; R.out2 = a.w.db ? a.w.db : a.w.dbs;
                R.slab_elems = a.w.p.slab_stride;
                R.n2 = a.w.extra / 4; R.nsplit = a.nsplit; R.ntap = a.w.ntap;
                R.CsOut = a.w.p.CsOut; R.cs_off = a.w.p.cs_off; R.cs_cnt = a.w.cs_cnt; R.CfOut = a.w.p.CfOut;
                R.block_begin = rb;
                rb += ew_reduce_job_blocks(R);
                wsp += (size_t)a.nsplit * a.w.p.slab_stride;
            }
        }
        if ((size_t)(wsp - ws) > rsu_wgrad_group_ws_floats()) return RSU_ENOMEM;
        // deal the units to the workgroups: longest first, each to the workgroup with the least work so far (ties: lowest id), then
        // every workgroup's list in the order it received them (long units first)
        std::stable_sort(ulist.begin(), ulist.end(), [](const U& a, const U& b) { return a.len > b.len; });
        const int nwg = (int)std::min<size_t>((size_t)ncu, ulist.size());
        std::vector<double> load((size_t)nwg, 0.0);
        std::vector<std::vector<unsigned>> mine((size_t)nwg);
        for (const U& u : ulist) {
            int m = 0;
            for (int c = 1; c < nwg; ++c)
                if (load[c] < load[m]) m = c;
            load[m] += u.len;
            mine[m].push_back(u.code);
        }
        int pos = 0;
        for (int b = 0; b < nwg; ++b) {
            T.g.wg_first[b] = pos;
            for (unsigned c : mine[b]) T.g.unit[pos++] = c;
        }
        for (int b = nwg; b <= 256; ++b) T.g.wg_first[b] = pos;
        T.g.njobs = n;
        T.g.nwg = nwg;
        T.nwg_total = nwg;
        T.red_blocks = rb;
        if (env_int("RSU_PLAN_DEBUG", 0)) {
            fprintf(stderr, "[plan wg group] %d jobs, %d units on %d CUs, makespan %.1f tile-steps (ideal %.1f), %d reduce jobs, %.1f MB of slabs, %d single launches\n",
                    n, (int)ulist.size(), ncu, best, W / ncu, T.nred, (double)(wsp - ws) * 4e-6, T.nsingle);
            for (int k = 0; k < n; ++k)
                fprintf(stderr, "   job %2d: family %2d blocks %3d x split %3d, %5d tiles, unit %.1f\n", k, T.g.job[k].family, T.g.job[k].gx * T.g.job[k].gy,
                        T.g.job[k].gz, T.g.job[k].p.ntiles_total, it[order[k]].unit);
        }
    }
    return RSU_OK;
}
extern "C" int rsu_wgrad_group_run(const void* host_table, const void* dev_table, rsu_stream_t stream) {
    if (!host_table || !dev_table) return RSU_EINVAL;
    const WgGroupTable& T = *(const WgGroupTable*)host_table;
    if (T.magic != WG_TABLE_MAGIC) return RSU_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (T.g.njobs > 0) {
        HIP_CHECK_RET(igemm_wg_group_launch((const IgWgGroupParams*)dev_table, T.nwg_total, st));
        if (T.nred > 0)
            HIP_CHECK_RET(ew_reduce_slabs_many((const ReduceJob*)((const char*)dev_table + offsetof(WgGroupTable, red)), T.nred, T.red_blocks, st));
    }
    for (int i = 0; i < T.nsingle; ++i) {   // (stream order: the group's slabs have been reduced before these reuse the workspace)
        const rsu_wgrad_job_t& j = T.single[i];
        int rc;
        if (j.kind == RSU_WGRAD_CONV3X3)
            rc = rsu_conv2d_bwd_weight(&j.src, j.dz, j.dw, j.db, T.ws, T.N, j.Ho, j.Wo, j.Cin_total, j.ci_off, j.Cout, j.dil, T.ncu, stream);
        else
            rc = rsu_convT2x2_bwd_weight(j.src.ptr, j.dz, j.dw, j.db, T.ws, T.N, j.src.H, j.src.W, j.src.C, j.Cout, T.ncu, stream);
        if (rc != RSU_OK) return rc;
    }
    return RSU_OK;
}

extern "C" size_t rsu_bias_grad_ws_floats(long npix, int C) { return (size_t)ew_colsum_blocks(npix, C) * C; }
extern "C" int rsu_bias_grad(const void* dz, float* db, float* ws, long npix, int C, rsu_stream_t stream) {
    if (!dz || !db || !ws || C < 8 || C % 8 || C > 2048 || 256 % (C / 8) || npix < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_colsum(dz, db, ws, npix, C, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// pool, head, optimizer, tiler
// ---------------------------------------------------------------------------------------------
extern "C" int rsu_maxpool2x2_fwd_code(const void* x, void* y, void* code, int N, int H, int W, int C, float keep, unsigned key,
                                       rsu_stream_t stream) {
    if (!x || !y || C % 8 || H < 2 || W < 2 || !keep_ok(keep)) return RSU_EINVAL;
    if (code && ((H | W) & 1)) return RSU_EINVAL;
    if (keep < 1.f && (long)N * (H / 2) * (W / 2) * C > 0xffffffffL) return RSU_EINVAL;
    HIP_CHECK_RET(ew_maxpool_fwd(x, y, code, N, H, W, C, keep, key, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_maxpool2x2_fwd(const void* x, void* y, int N, int H, int W, int C, float keep, unsigned key, rsu_stream_t stream) {
    return rsu_maxpool2x2_fwd_code(x, y, nullptr, N, H, W, C, keep, key, stream);
}
extern "C" int rsu_pool_skip_relu_bwd_code(const void* y_act, const void* code, const void* dpool, const void* dskip, void* dz, int N, int H,
                                           int W, int C, int Hs, int Ws, float keep, unsigned key, rsu_stream_t stream) {
    if ((!y_act && !code) || !dz || C % 8 || !keep_ok(keep)) return RSU_EINVAL;
    if (code && (((H | W) & 1) || H < 2 || W < 2)) return RSU_EINVAL;   // the code tensor describes whole 2x2 windows
    if (dskip && (Hs > H || Ws > W || Hs < 1 || Ws < 1)) return RSU_EINVAL;
    HIP_CHECK_RET(ew_pool_skip_relu_bwd(y_act, code, dpool, dskip, dz, N, H, W, C, dskip ? Hs : 0, dskip ? Ws : 0, keep, key, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_pool_skip_relu_bwd(const void* y_act, const void* dpool, const void* dskip, void* dz, int N, int H, int W, int C,
                                      int Hs, int Ws, float keep, unsigned key, rsu_stream_t stream) {
    if (!y_act) return RSU_EINVAL;
    return rsu_pool_skip_relu_bwd_code(y_act, nullptr, dpool, dskip, dz, N, H, W, C, Hs, Ws, keep, key, stream);
}
static bool head_c_ok(int C) { return C >= 8 && C <= 512 && (C % 8) == 0 && ((C / 8) & (C / 8 - 1)) == 0; }
extern "C" int rsu_color_adjust_bwd(const float* gx, const float* w1, float* dW0, float* db0, int Cout, float scale, int accumulate,
                                    rsu_stream_t stream) {
    if (!gx || !w1 || !dW0 || !db0 || Cout < 1 || !(scale > 0.f)) return RSU_EINVAL;
    HIP_CHECK_RET(ew_color_adjust_bwd(gx, w1, dW0, db0, Cout, scale, accumulate, (hipStream_t)stream));
    return RSU_OK;
}

extern "C" int rsu_head_fwd(const void* act, const float* w, const float* b, float* prob, float* logits, long npix, int C,
                            rsu_stream_t stream) {
    if (!act || !w || !b || !prob || !head_c_ok(C) || npix < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_head(false, act, w, b, nullptr, prob, logits, nullptr, nullptr, nullptr, nullptr, nullptr, npix, C, 0.f, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" size_t rsu_head_ws_floats(long npix, int C) { return (size_t)ew_head_blocks(npix, C) * (2 * C + 3); }
extern "C" int rsu_head_fwd_bwd(const void* act, const float* w, const float* b, const int64_t* labels, float* prob, float* loss_sum,
                                void* dact, float* dw, float* db, float* ws, long npix, int C, float inv_count, rsu_stream_t stream) {
    if (!act || !w || !b || !labels || !prob || !loss_sum || !dact || !dw || !db || !ws || !head_c_ok(C) || npix < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_head(true, act, w, b, labels, prob, nullptr, dact, dw, db, loss_sum, ws, npix, C, inv_count, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_momentum_step(float* w, float* acc, const float* g, float lr, float mu, float gscale, long n, rsu_stream_t stream) {
    if (!w || !acc || !g || n < 1) return RSU_EINVAL;
    if (((uintptr_t)w | (uintptr_t)acc | (uintptr_t)g) & 15) return RSU_EINVAL;
    HIP_CHECK_RET(ew_momentum(w, acc, g, lr, mu, gscale, n, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_extract_tiles(const float* imgs, float* tiles, int nimg, int H, int S, int P, int stride, long t0, long ntiles,
                                 rsu_stream_t stream) {
    // images.py:60-61 asserts + tf_aerial_images.py:288-293 geometry
    if (!imgs || !tiles || P > S || (S - P) % 2 || P > H || stride < 1 || (H - P) % stride || (S - P) / 2 > H) return RSU_EINVAL;
    const int pps = (H - P) / stride + 1;
    if (t0 < 0 || ntiles < 1 || t0 + ntiles > (long)nimg * pps * pps) return RSU_EINVAL;
    HIP_CHECK_RET(ew_extract_tiles(imgs, tiles, H, S, P, stride, pps, t0, ntiles, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_overlap_add(const float* prob, float* acc, float* hits, int nimg, int H, int P, int stride, long t0, long ntiles,
                               rsu_stream_t stream) {
    if (!prob || !acc || !hits || P > H || stride < 1 || (H - P) % stride) return RSU_EINVAL;
    const int pps = (H - P) / stride + 1;
    if (t0 < 0 || ntiles < 1 || t0 + ntiles > (long)nimg * pps * pps) return RSU_EINVAL;
    HIP_CHECK_RET(ew_overlap_add(prob, acc, hits, nimg, H, P, stride, pps, t0, ntiles, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_overlap_finish(const float* acc, const float* hits, float* out, long n, rsu_stream_t stream) {
    if (!acc || !hits || !out || n < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_overlap_finish(acc, hits, out, n, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// post-processing wire format + metrics counters
// ---------------------------------------------------------------------------------------------
extern "C" int rsu_quantize_mask(const float* masks, float* out, int nimg, int S, int patch_size, float threshold, rsu_stream_t stream) {
    if (!masks || !out || nimg < 1 || S < 1 || patch_size < 1 || patch_size > 64) return RSU_EINVAL;
    HIP_CHECK_RET(ew_block_label(masks, out, nullptr, nimg, S, patch_size, threshold, 0, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_labels_for_patches(const float* masks, int64_t* labels, int nimg, int S, int patch_size, float threshold,
                                      rsu_stream_t stream) {
    if (!masks || !labels || nimg < 1 || S < 1 || patch_size < 1 || patch_size > 64 || S % patch_size) return RSU_EINVAL;
    HIP_CHECK_RET(ew_block_label(masks, nullptr, labels, nimg, S, patch_size, threshold, 1, (hipStream_t)stream));
    return RSU_OK;
}
extern "C" int rsu_confusion_counts(const int64_t* predictions, const int64_t* labels, long n, unsigned long long* counts,
                                    rsu_stream_t stream) {
    if (!predictions || !labels || !counts || n < 1) return RSU_EINVAL;
    HIP_CHECK_RET(ew_confusion(predictions, labels, n, counts, (hipStream_t)stream));
    return RSU_OK;
}

// ---------------------------------------------------------------------------------------------
// rsu_plan: the static shape table of unet.forward (src/unet.py:12-97) for a (num_layers, root_size, patch_size, dilated) network
// ---------------------------------------------------------------------------------------------
extern "C" int rsu_plan(int num_layers, int root_size, int patch_size, int dilated, int batch, rsu_plan_row_t* rows, int capacity,
                        int* nrows, rsu_plan_totals_t* totals) {
    if (num_layers < 1 || root_size < 8 || root_size % 8 || patch_size < 1 || batch < 1 || !nrows) return RSU_EINVAL;
    int S = 0;
    if (rsu_input_size_needed(patch_size, num_layers, &S) != RSU_OK) return RSU_EINVAL;
    const int L = num_layers;
    int n = 0;
    long params = 0, act_elems = 0;
    size_t ws = 0;
    auto upd_ws = [&](size_t w) { if (w > ws) ws = w; };
    auto add = [&](int kind, int level, int Hin, int Cin, int Cout, int Hout, int dil, int nsrc) {
        if (rows && n < capacity) {
            rsu_plan_row_t r;
            r.kind = kind; r.level = level; r.Hin = Hin; r.Win = Hin; r.Cin = Cin; r.Cout = Cout; r.Hout = Hout; r.Wout = Hout;
            r.dilation = dil; r.nsrc = nsrc;
            rows[n] = r;
        }
        ++n;
        act_elems += (long)batch * Hout * Hout * Cout;
    };
    auto conv_params = [&](int k, int cin, int cout) { params += (long)k * k * cin * cout + cout; };
    add(RSU_OP_COLOR_ADJUST, 0, S, 3, 3, S, 1, 1);
    conv_params(1, 3, 3);
    int h = S, nf = root_size, cin = 3;
    for (int i = 0; i < L; ++i) {
        if (dilated) {  // unet.py:32-39: the dilated twin block (built at every level; the level L-1 pair is never consumed)
            conv_params(3, cin, nf);
            conv_params(3, nf, nf);
            if (i < L - 1) {
                add(RSU_OP_CONV3X3, i, h, cin, nf, h - 4, 2, 1);
                add(RSU_OP_CONV3X3, i, h - 4, nf, nf, h - 8, 2, 1);
                if (cin != 3) upd_ws(rsu_conv2d_bwd_weight_ws_floats(cin, cin, nf));
                upd_ws(rsu_conv2d_bwd_weight_ws_floats(nf, nf, nf));
            }
        }
        add(RSU_OP_CONV3X3, i, h, cin, nf, h - 2, 1, 1);
        add(RSU_OP_CONV3X3, i, h - 2, nf, nf, h - 4, 1, 1);
        conv_params(3, cin, nf);
        conv_params(3, nf, nf);
        if (cin != 3) upd_ws(rsu_conv2d_bwd_weight_ws_floats(cin, cin, nf)); else upd_ws(rsu_conv_first_bwd_ws_floats(nf));
        upd_ws(rsu_conv2d_bwd_weight_ws_floats(nf, nf, nf));
        if (i < L - 1) {
            if ((h - 4) % 2) return RSU_EINVAL;
            add(RSU_OP_MAXPOOL, i, h - 4, nf, nf, (h - 4) / 2, 1, 1);
            h = (h - 4) / 2;
            cin = nf;
            nf *= 2;
        }
    }
    h -= 4;
    for (int i = 0; i < L - 1; ++i) {
        nf /= 2;
        const int j = L + i, nsrc = dilated ? 3 : 2;
        add(RSU_OP_CONVT2X2, j, h, 2 * nf, nf, 2 * h, 1, 1);
        params += (long)4 * nf * 2 * nf + nf;
        upd_ws(rsu_convT2x2_bwd_weight_ws_floats(2 * nf, nf));
        h *= 2;
        add(RSU_OP_CONV3X3, j, h, nsrc * nf, nf, h - 2, 1, nsrc);  // concat [skip, (dilated skip), up] never materialised
        add(RSU_OP_CONV3X3, j, h - 2, nf, nf, h - 4, 1, 1);
        conv_params(3, nsrc * nf, nf);
        conv_params(3, nf, nf);
        upd_ws(rsu_conv2d_bwd_weight_ws_floats(nsrc * nf, nf, nf));
        upd_ws(rsu_conv2d_bwd_weight_ws_floats(nf, nf, nf));
        h -= 4;
    }
    if (h != patch_size) return RSU_EINVAL;
    add(RSU_OP_HEAD, 2 * L - 1, h, nf, 2, h, 1, 1);
    conv_params(1, nf, 2);
    upd_ws(rsu_head_ws_floats((long)batch * h * h, nf));
    *nrows = n;
    if (totals) {
        totals->input_size = S;
        totals->num_params = params;
        totals->activation_elems = act_elems;
        totals->workspace_floats = ws;
    }
    return (rows && n > capacity) ? RSU_ENOMEM : RSU_OK;
}
