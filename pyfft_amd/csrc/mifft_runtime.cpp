// C-ABI runtime shim + pass launchers of libmifft.so (see include/mifft.h for the contract and for
// the reference interfaces each entry point replaces).
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "../../include/mifft.h"
#include "mifft_internal.h"

namespace {

thread_local char g_err[512] = "";
// the library's only process-wide state: the DEFAULTS of the development switches (all zero in production) and a per-device
// compute-unit count.  The switches: a process default (mifft_debug_set_default) under a per-thread override (mifft_debug_set) -- a switch flipped
// by one thread never changes what another thread's plan launches
std::atomic<int> g_debug_default[MIFFT_DEBUG_KEYS];
thread_local int t_debug_value[MIFFT_DEBUG_KEYS];
thread_local unsigned t_debug_mask = 0;
struct DebugSwitches {
    int operator[](int key) const {
        return ((t_debug_mask >> key) & 1u) ? t_debug_value[key] : g_debug_default[key].load(std::memory_order_relaxed);
    }
};
const DebugSwitches g_debug;

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return set_err((int)e, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

bool is_pow2(long long v) { return v > 0 && (v & (v - 1)) == 0; }
// a * b * c (all >= 0) without overflow, or -1
long long mul3_checked(long long a, long long b, long long c) {
    long long ab, abc;
    if (__builtin_mul_overflow(a, b, &ab) || __builtin_mul_overflow(ab, c, &abc)) return -1;
    return abc;
}
// rows of the mixed-radix / Bluestein launchers: interleaved complex numbers, 8-byte (fp32) / 16-byte (fp64) aligned; an in-place call
// must keep every row where it is (a work-group stores its rows while others have not loaded theirs yet)
const char* check_rows(int precision, const void* in, const void* out, long long rows, long long n, long long stride_in, long long stride_out) {
    const uintptr_t mask = precision == MIFFT_F64 ? 15 : 7;
    if (((uintptr_t)in | (uintptr_t)out) & mask) return "data buffers must be aligned to one complex number";
    if (in == out && stride_in != stride_out) return "an in-place call needs equal row strides on both sides";
    const long long esz = precision == MIFFT_F64 ? 16 : 8;
    if (mul3_checked(rows, stride_in > stride_out ? stride_in : stride_out, esz) < 0) return "rows * stride overflows";
    (void)n;
    return nullptr;
}
int ilog2(long long v) {
    int r = 0;
    while (v > 1) {
        v >>= 1;
        ++r;
    }
    return r;
}

int validate(const mifft_pass* p) {
    if (!p) return set_err(MIFFT_E_INVALID, "null pass descriptor");
    if (p->kind != MIFFT_PASS_COL && p->kind != MIFFT_PASS_ROW && p->kind != MIFFT_PASS_ND) return set_err(MIFFT_E_INVALID, "bad pass kind %d", p->kind);
    if (p->precision != MIFFT_F32 && p->precision != MIFFT_F64) return set_err(MIFFT_E_INVALID, "bad precision %d", p->precision);
    if (p->layout != MIFFT_INTERLEAVED && p->layout != MIFFT_SPLIT) return set_err(MIFFT_E_INVALID, "bad layout %d", p->layout);
    if (!is_pow2(p->L) || (p->L < 2 && p->kind != MIFFT_PASS_ND)) return set_err(MIFFT_E_INVALID, "L=%d is not a power of two >= 2", p->L);
    if (p->outer < 0) return set_err(MIFFT_E_INVALID, "negative outer count");
    if (p->kind == MIFFT_PASS_ND) {
        if (!is_pow2(p->M) || !is_pow2(p->S)) return set_err(MIFFT_E_INVALID, "ND pass: y and z must be powers of two");
        const long long n = (long long)p->L * p->M * p->S;
        const bool both_interleaved = p->layout != MIFFT_SPLIT || ((p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && (p->flags & MIFFT_FLAG_DST_INTERLEAVED));
        // (planes on the input side at least: the tiled fixed-shape kernel takes planes -> planes and planes -> interleaved)
        const bool both_split = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED);
        // (interleaved shapes that exist out of place only: launch_nd refuses an in-place call, with its own message)
        if (n < 4 || (mifft_nd_shape_supported(p->precision, p->L, (int32_t)p->M, (int32_t)p->S,
                                               both_interleaved ? MIFFT_VARIANT_INTERLEAVED_ONLY : both_split ? MIFFT_VARIANT_SPLIT_ONLY : 0) != 0 &&
                      !(both_interleaved && mifft_nd_shape_supported(p->precision, p->L, (int32_t)p->M, (int32_t)p->S, MIFFT_VARIANT_OUT_OF_PLACE_ONLY) == 0) &&
                      !(both_split && !(p->flags & MIFFT_FLAG_DST_INTERLEAVED) &&
                        mifft_nd_shape_supported(p->precision, p->L, (int32_t)p->M, (int32_t)p->S, MIFFT_VARIANT_SPLIT_OUT_OF_PLACE) == 0)))
            return set_err(MIFFT_E_UNSUPPORTED, "ND pass: no kernel for %d x %lld x %lld (%lld points)", p->L, (long long)p->M, (long long)p->S, n);
        if ((p->L > 1 && !p->tw_L) || (p->M > 1 && !p->tw_lo) || (p->S > 1 && !p->tw_hi)) return set_err(MIFFT_E_INVALID, "ND pass: twiddle table missing");
        return 0;
    }
    if (!p->tw_L) return set_err(MIFFT_E_INVALID, "tw_L table missing");
    if (p->kind == MIFFT_PASS_COL) {
        if (!is_pow2(p->M) || !is_pow2(p->S)) return set_err(MIFFT_E_INVALID, "M and S must be powers of two");
        if (p->M * p->S < 2) return set_err(MIFFT_E_INVALID, "a COL pass needs M*S >= 2 (use a ROW pass)");
        if ((long long)p->L * p->M > (1ll << 30)) return set_err(MIFFT_E_INVALID, "axis longer than 2^30");
        if (p->M > 1) {
            if (!p->tw_lo || !p->tw_hi) return set_err(MIFFT_E_INVALID, "inter-pass twiddle tables missing (M > 1)");
            if (p->tw_shift < 0 || p->tw_shift > 30) return set_err(MIFFT_E_INVALID, "bad tw_shift");
        }
    } else {
        if (p->M != 1 || p->S != 1) return set_err(MIFFT_E_INVALID, "a ROW pass has M == S == 1");
    }
    return 0;
}

// TileArgs.nt from the MIFFT_FLAG_STREAM_* hints: bit 0 non-temporal loads, bit 1 non-temporal stores, bit 2 write-through stores
// (development switch MIFFT_DEBUG_STORE overrides the store side for A/B measurements)
int stream_policy(int flags) {
    int nt = ((flags & MIFFT_FLAG_STREAM_SRC) ? 1 : 0) | ((flags & MIFFT_FLAG_STREAM_DST) ? 2 : 0);
    if (flags & MIFFT_FLAG_WRITE_THROUGH) nt = (nt & 1) | 4;
    if (g_debug[MIFFT_DEBUG_STORE] == 1) nt = (nt & 1) | 2;
    else if (g_debug[MIFFT_DEBUG_STORE] == 2) nt = (nt & 1) | 4;
    else if (g_debug[MIFFT_DEBUG_STORE] == 3) nt = nt & 1;
    return nt;
}

void fill_args(const mifft_pass* p, const void* in0, const void* in1, void* out0, void* out1, mifft::TileArgs* pa) {
    mifft::TileArgs& a = *pa;
    const bool split = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED);
    const bool split_out = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_DST_INTERLEAVED);
    a.in0 = in0;
    a.in1 = in1;
    a.out0 = out0;
    a.out1 = out1;
    a.tw_L = p->tw_L;
    a.tw_lo = p->tw_lo;
    a.tw_hi = p->tw_hi;
    a.ostride_in = p->outer_stride_in;
    a.ostride_out = p->outer_stride_out;
    if (p->kind == MIFFT_PASS_COL) {
        a.total = p->outer * p->M * p->S;
        a.logMS = ilog2(p->M * p->S);
        a.logS = ilog2(p->S);
    } else {
        a.total = p->outer;
        a.logMS = 0;
        a.logS = 0;
    }
    a.tw_shift = p->tw_shift;
    a.split = split ? 1 : 0;
    a.split_out = split_out ? 1 : 0;
    a.inverse = p->inverse ? 1 : 0;
    a.has_tw = (p->kind == MIFFT_PASS_COL && p->M > 1) ? 1 : 0;
    a.nt = stream_policy(p->flags);
    a.scale = p->scale;
}

// radix list of one axis for the in-LDS N-D kernel (each radix <= maxr, a power of two)
int nd_radices(int L, int maxr, int* out) {
    int n = 0;
    while (L > 1) {
        int r = maxr;
        while (r > L) r >>= 1;
        // avoid a trailing radix 2 after big radices: prefer e.g. 32 = 8 * 4 over 16 * 2
        if (L > r && L / r == 2 && r >= 8) r >>= 1;
        out[n++] = r;
        L /= r;
    }
    return n;
}

// compute units of the CURRENT device (cached per device index: a process may drive several different devices)
int current_cus() {
    static std::atomic<int> cached[64];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64 && (cus = cached[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 256;
    if (dev >= 0 && dev < 64) cached[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// grid cap of the grid-stride wave kernels: 8 work-groups of 4 waves per CU
int wave_max_blocks() { return current_cus() * 8; }

// The statically dealt sequential work list (lag == 0) is deadlock-free only while every work-group of the launch is resident:
// the grid is capped at `per_cu` work-groups per compute unit (what the kernel's registers and LDS allow), whatever the caller asked for
int resident_grid(int grid, int per_cu) {
    const int cus = current_cus();
    const int cap = per_cu * (cus > 0 ? cus : 1);
    return grid > cap ? cap : grid;
}

long long wave_bytes(const mifft_pass* p, const mifft::TileArgs* a) {   // bytes of one side of a dense ROW pass
    return a->total * p->L * (p->precision == MIFFT_F64 ? 16ll : 8ll);
}

// which form a fixed N-D shape runs: two work-groups per transform (fft_nd2z.hip) for the one-tile-per-CU shapes wherever that kernel
// exists (it is only instantiated where it measured faster, at 32 MiB and at 1 GiB per side), for the two-per-CU shapes in SMALL launches
// only -- the plan marks those with MIFFT_FLAG_WRITE_THROUGH (up to half the last-level cache per side).  A/B: MIFFT_DEBUG_ALT_ROWS = 6 never
bool nd2z_preferred(bool f64, int x, int y, int z, bool small_launch) {
    return g_debug[MIFFT_DEBUG_ALT_ROWS] != 6 && mifft_nd2z(f64 ? 1 : 0, x, y, z, nullptr, nullptr, small_launch ? 1 : 2) == 0;
}

int launch_nd(const mifft_pass* p, const void* in0, const void* in1, void* out0, void* out1, hipStream_t s) {
    // fixed-shape kernels (fft_nd2.hpp) for the common shapes, interleaved on both sides
    // the run-time-shaped kernel only: the development switch, or variant 1 of the pass (the plan asks for it where that kernel measured
    // faster than the shape's fixed instance: pyfft_amd/tuning_gfx950.json, "nd_generic")
    const bool no_nd2 = g_debug[MIFFT_DEBUG_NO_ND2] != 0 || p->variant == 1;
    // the (16, 16) fp32 plane, interleaved: wave-autonomous kernel (no LDS, DPP exchange; csrc/fft_wave.hpp)
    // -- measured slower than the fixed-shape LDS kernel at the reference's 32 MiB buffer (0.62 vs 0.68 of the roofline) and
    // equal at 128 MiB (profiles/r02_h_small_n.log), so it only runs on request (MIFFT_DEBUG_FORCE_WAVE; parity-tested)
    if (p->precision == MIFFT_F32 && p->L == 16 && p->M == 16 && p->S == 1 && g_debug[MIFFT_DEBUG_FORCE_WAVE] &&
        (p->layout != MIFFT_SPLIT || ((p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && (p->flags & MIFFT_FLAG_DST_INTERLEAVED)))) {
        mifft::WaveArgs w;
        w.in = in0; w.out = out0;
        w.pieces = p->outer * 128;
        w.inverse = p->inverse ? 1 : 0;
        w.nt = stream_policy(p->flags);
        w.scale = p->scale;
        const int rc = mifft_wave_16x16_launch(&w, wave_max_blocks(), s);
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    const bool f64nd = p->precision == MIFFT_F64;
    const int have_nd2 = f64nd ? mifft_nd2_f64_supported((int)p->L, (int)p->M, (int)p->S)
                               : mifft_nd2_f32_supported((int)p->L, (int)p->M, (int)p->S);
    const bool inter_both = p->layout != MIFFT_SPLIT || ((p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && (p->flags & MIFFT_FLAG_DST_INTERLEAVED));
    // Several work-groups per transform (fft_nd2z.hpp, round 5; out of place only): the one-tile-per-CU shapes always, the two-per-CU
    // shapes in small launches, and shapes of FOUR two-per-CU tiles, which have no other one-launch kernel (the plan only builds such a
    // pass for its out-of-place executes)
    if (inter_both && !no_nd2 && in0 != out0 &&
        (have_nd2 == 0 ? nd2z_preferred(f64nd, (int)p->L, (int)p->M, (int)p->S, (p->flags & MIFFT_FLAG_WRITE_THROUGH) != 0)
                       : mifft_nd2z(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, nullptr, nullptr, 1) == 0)) {
        mifft::TileArgs t;
        memset(&t, 0, sizeof(t));
        t.in0 = in0; t.out0 = out0;
        t.tw_L = p->tw_L; t.tw_lo = p->tw_lo; t.tw_hi = p->tw_hi;
        t.total = p->outer * p->L * p->M * p->S;
        t.inverse = p->inverse ? 1 : 0;
        t.scale = p->scale;
        t.nt = stream_policy(p->flags);
        const int rz = mifft_nd2z(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, &t, s, 0);
        if (rz == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rz != 0) return hip_check((hipError_t)rz, "kernel launch");
        return 0;
    }
    if (have_nd2 == 0 &&
        (p->layout != MIFFT_SPLIT || ((p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && (p->flags & MIFFT_FLAG_DST_INTERLEAVED))) &&
        !no_nd2) {
        mifft::TileArgs t;
        memset(&t, 0, sizeof(t));
        t.in0 = in0; t.out0 = out0;
        t.tw_L = p->tw_L; t.tw_lo = p->tw_lo; t.tw_hi = p->tw_hi;
        t.total = p->outer * p->L * p->M * p->S;
        t.inverse = p->inverse ? 1 : 0;
        t.scale = p->scale;
        t.nt = stream_policy(p->flags);
        const int rc = f64nd ? mifft_nd2_f64_launch((int)p->L, (int)p->M, (int)p->S, &t, s)
                             : mifft_nd2_f32_launch((int)p->L, (int)p->M, (int)p->S, &t, s);
        if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    // split-complex planes on both sides (a single-pass N-D plan): the fixed-shape stage chain exists for planes in its TILED form
    // (fft_nd2t.hpp, second batch of round 4) -- a dense batch is the tiling with one tile per "parent".  (128, 128) planes at 1 GiB:
    // 0.471 on the run-time-shaped kernel below -> 0.653 (interleaved fixed-shape kernel 0.700), (64, 64) 0.535 -> 0.597, fp64 16^3
    // 0.510 -> 0.584; shapes whose x rows are shorter than 128 bytes per plane stay below -- fp32 (16, 16) 0.717 against 0.460, 16^3 0.674
    // against 0.475: their tiles move scalars over short runs (profiles/r04_at_rows_split.log)
    // (planes in, interleaved out -- the plane pass of a split-complex multi-pass plan -- likewise: fft_nd2t_split_in.hip)
    // Round 6: planes on BOTH sides of a published shape: the dense kernel that moves 16 bytes per lane and plane (fft_nd2p.hpp); the tiled
    // kernel below moves one scalar per lane and plane through the tiling's address arithmetic (0.69-0.86 of the interleaved twin at 1 GiB).
    // MIFFT_DEBUG_ALT_ROWS = 7: the tiled kernel (A/B)
    // ... and its one-tile-per-CU shapes as two half-size work-groups per transform, out of place (fft_nd2zp.hpp; MIFFT_DEBUG_ALT_ROWS = 6:
    // never several work-groups per transform, as for interleaved data)
    if (p->layout == MIFFT_SPLIT && !(p->flags & (MIFFT_FLAG_SRC_INTERLEAVED | MIFFT_FLAG_DST_INTERLEAVED)) && !no_nd2 && in1 && out1 &&
        in0 != out0 && in1 != out1 && g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && g_debug[MIFFT_DEBUG_ALT_ROWS] != 7 &&
        g_debug[MIFFT_DEBUG_ALT_ROWS] != 6 && mifft_nd2zp(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, nullptr, nullptr, 1) == 0) {
        mifft::TileArgs t;
        memset(&t, 0, sizeof(t));
        t.in0 = in0; t.in1 = in1; t.out0 = out0; t.out1 = out1;
        t.split = 1; t.split_out = 1;
        t.tw_L = p->tw_L; t.tw_lo = p->tw_lo; t.tw_hi = p->tw_hi;
        t.total = p->outer * p->L * p->M * p->S;
        t.inverse = p->inverse ? 1 : 0;
        t.scale = p->scale;
        t.nt = stream_policy(p->flags);
        const int rc = mifft_nd2zp(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, &t, s, 0);
        if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    if (p->layout == MIFFT_SPLIT && !(p->flags & (MIFFT_FLAG_SRC_INTERLEAVED | MIFFT_FLAG_DST_INTERLEAVED)) && !no_nd2 && in1 && out1 &&
        g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && g_debug[MIFFT_DEBUG_ALT_ROWS] != 7 &&
        mifft_nd2p(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, nullptr, nullptr, (p->flags & MIFFT_FLAG_WRITE_THROUGH) ? 2 : 1) == 0) {
        mifft::TileArgs t;
        memset(&t, 0, sizeof(t));
        t.in0 = in0; t.in1 = in1; t.out0 = out0; t.out1 = out1;
        t.split = 1; t.split_out = 1;
        t.tw_L = p->tw_L; t.tw_lo = p->tw_lo; t.tw_hi = p->tw_hi;
        t.total = p->outer * p->L * p->M * p->S;
        t.inverse = p->inverse ? 1 : 0;
        t.scale = p->scale;
        t.nt = stream_policy(p->flags);
        const int rc = mifft_nd2p(f64nd ? 1 : 0, (int)p->L, (int)p->M, (int)p->S, &t, s, 0);
        if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    const bool planes_in_only = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && (p->flags & MIFFT_FLAG_DST_INTERLEAVED);
    if (p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED) && !no_nd2 && in1 && (out1 || planes_in_only) &&
        g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && p->L * (f64nd ? 8 : 4) >= 128 &&
        mifft_nd2t_split(f64nd, (int)p->L, (int)p->M, (int)p->S, nullptr, nullptr, nullptr, 1) == 0) {
        mifft::TileArgs t;
        memset(&t, 0, sizeof(t));
        t.in0 = in0; t.in1 = in1; t.out0 = out0; t.out1 = planes_in_only ? nullptr : out1;
        t.split = 1;
        t.split_out = planes_in_only ? 0 : 1;
        t.nt = stream_policy(p->flags) & 4;      // (the kernel knows the write-through form of the small launches only)
        t.tw_L = p->tw_L; t.tw_lo = p->tw_lo; t.tw_hi = p->tw_hi;
        t.inverse = p->inverse ? 1 : 0;
        t.scale = p->scale;
        mifft::TiledGeom g;
        g.pitch_y = p->L; g.pitch_z = (long long)p->L * p->M; g.parent = (long long)p->L * p->M * p->S; g.tiles = p->outer;
        g.cx = g.cy = g.cz = 1;
        const int rc = (planes_in_only ? mifft_nd2t_split_in : mifft_nd2t_split)(f64nd, (int)p->L, (int)p->M, (int)p->S, &t, &g, s, 0);
        if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    if ((long long)p->L * p->M * p->S > mifft_nd_max_points(f64nd))
        return set_err(MIFFT_E_UNSUPPORTED, "ND pass %d x %d x %d: this shape has a one-launch kernel out of place only, for interleaved data or planes on both sides "
                       "(several work-groups per transform, mifft_nd_shape_supported with MIFFT_VARIANT_OUT_OF_PLACE_ONLY / _SPLIT_OUT_OF_PLACE)",
                       (int)p->S, (int)p->M, (int)p->L);
    mifft::NdArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1;
    a.tw[0] = p->tw_L; a.tw[1] = p->tw_lo; a.tw[2] = p->tw_hi;
    const long long dims[3] = {p->L, p->M, p->S};
    a.total = p->outer * dims[0] * dims[1] * dims[2];
    const bool f64 = p->precision == MIFFT_F64;
    int logs = 0, ns = 0;
    for (int ax = 0; ax < 3; ++ax) {
        a.logL[ax] = ilog2(dims[ax]);
        a.logS[ax] = logs;
        logs += a.logL[ax];
        int rad[16];
        const int nr = nd_radices((int)dims[ax], f64 ? 8 : 16, rad);
        int logNs = 0;
        for (int i = 0; i < nr; ++i) {
            if (ns >= mifft::kNdMaxStages) return set_err(MIFFT_E_UNSUPPORTED, "ND pass: too many stages");
            a.st_axis[ns] = (unsigned char)ax;
            a.st_radix[ns] = (unsigned char)rad[i];
            a.st_logNs[ns] = (unsigned char)logNs;
            logNs += ilog2(rad[i]);
            ++ns;
        }
    }
    a.nstages = ns;
    a.split = (p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED)) ? 1 : 0;
    a.split_out = (p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_DST_INTERLEAVED)) ? 1 : 0;
    a.inverse = p->inverse ? 1 : 0;
    a.scale = p->scale;
    // register edge: the last stage of all writes runs of (L / radix) * S points; straight to HBM when that is >= 128 bytes
    if (ns >= 1) {
        const int axn = a.st_axis[ns - 1];
        const long long run_out = ((dims[axn] / a.st_radix[ns - 1]) << a.logS[axn]) * (f64 ? 16 : 8);
        a.edge_out = (!a.split_out && run_out >= 128) ? 1 : 0;
    }
    // small launches: write-through stores of the result (the store phase -- planes always go through it; an interleaved result that
    // leaves by the register edge keeps plain stores).  MIFFT_NARROW_TILES = 1: off (A/B)
    a.wt = ((stream_policy(p->flags) & 4) && g_debug[MIFFT_DEBUG_NARROW_TILES] != 1) ? 1 : 0;
    const int rc = mifft_nd_launch(f64 ? 1 : 0, dims[0] * dims[1] * dims[2], &a, s);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int dispatch(const mifft_pass* p, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    const int tr = (p->kind == MIFFT_PASS_COL && p->S == 1) ? 1 : 0;
    int rc;
    // short dense interleaved rows: wave-autonomous kernel (no LDS, DPP exchange; csrc/fft_wave.hpp)
    // up to 128 MiB per side (both sides then fit the 256 MiB Infinity Cache: 0.72-0.84 of the roofline against 0.57-0.69
    // for the LDS-staged kernels at the reference's 32 MiB buffer, 0.90-0.99 against 0.83-0.90 at 128 MiB); larger
    // buffers are HBM-bound and the LDS kernels are 0-5 points ahead (profiles/r02_h_small_n.log)
    if (!query_only && p->kind == MIFFT_PASS_ROW && a && !a->split && !a->split_out && a->ostride_in == p->L &&
        a->ostride_out == p->L && !g_debug[MIFFT_DEBUG_NO_WAVE] && mifft_wave_supported(p->precision == MIFFT_F64, p->L) == 0 &&
        (g_debug[MIFFT_DEBUG_FORCE_WAVE] ||
         (wave_bytes(p, a) <= (128ll << 20) && !(p->precision == MIFFT_F32 && p->L == 32 && wave_bytes(p, a) < (64ll << 20))))) {
        mifft::WaveArgs w;
        w.in = a->in0; w.out = a->out0;
        w.pieces = a->total * p->L * (p->precision == MIFFT_F64 ? 16 : 8) / 16;
        w.inverse = a->inverse;
        w.nt = a->nt;
        w.scale = a->scale;
        rc = mifft_wave_launch(p->precision == MIFFT_F64, p->L, &w, wave_max_blocks(), s);
        if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
        return 0;
    }
    if (p->kind == MIFFT_PASS_COL)
        rc = p->precision == MIFFT_F32 ? mifft_dispatch_col_f32(p->L, tr, p->variant, a, s, query_only)
                                       : mifft_dispatch_col_f64(p->L, tr, p->variant, a, s, query_only);
    else
        rc = p->precision == MIFFT_F32 ? mifft_dispatch_row_f32(p->L, p->variant, a, s, query_only)
                                       : mifft_dispatch_row_f64(p->L, p->variant, a, s, query_only);
    if (rc == MIFFT_E_UNSUPPORTED)
        return set_err(rc, "no compiled kernel for kind=%d precision=%d L=%d variant=%d", p->kind, p->precision, p->L, p->variant);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

// ---- pass pairs (fft_pair.hpp) -------------------------------------------------------------------------------------
// kind 0 = XY (ROW x + COL y R0), kind 1 = YZ (COL y R1 + COL z); keys as in mifft_pair_f64; *split = the user-facing side of
// the launch is two scalar planes (XY: its input, YZ: its output; the side between the two launches is always interleaved).
// Returns 0 and fills kind / keys / split, or MIFFT_E_UNSUPPORTED when (p0, p1) is not a pair shape.
int classify_pair(const mifft_pass* p0, const mifft_pass* p1, int* kind, int key[3], int* split) {
    if (!p0 || !p1) return MIFFT_E_INVALID;
    if (p0->precision != p1->precision || p0->inverse != p1->inverse || p0->layout != p1->layout) return MIFFT_E_UNSUPPORTED;
    const bool inter_in = p0->layout != MIFFT_SPLIT || (p0->flags & MIFFT_FLAG_SRC_INTERLEAVED);
    const bool inter_out = p1->layout != MIFFT_SPLIT || (p1->flags & MIFFT_FLAG_DST_INTERLEAVED);
    if (p0->kind == MIFFT_PASS_ROW && p1->kind == MIFFT_PASS_COL && p1->M > 1 && p1->S == p0->L) {
        const long long plane = (long long)p0->L * p1->L * p1->M;            // nx * ny
        if (!inter_out || p1->outer_stride_in != plane || p1->outer_stride_out != plane || p0->outer != p1->outer * p1->L * p1->M ||
            p0->outer_stride_in != p0->L || p0->outer_stride_out != p0->L || p1->M > (1 << 20))
            return MIFFT_E_UNSUPPORTED;
        *kind = 0;
        *split = inter_in ? 0 : 1;
        key[0] = p0->L; key[1] = p1->L; key[2] = (int)p1->M;
        return 0;
    }
    if (p0->kind == MIFFT_PASS_COL && p1->kind == MIFFT_PASS_COL && p0->M == 1 && p1->M == 1 && p1->S == p0->S * p0->L) {
        const long long xform = p1->S * p1->L;                                // nx * ny * nz
        if (!inter_in || p1->outer_stride_in != xform || p1->outer_stride_out != xform || p0->outer != p1->outer * p1->L ||
            p0->outer_stride_in != p1->S || p0->outer_stride_out != p1->S || p0->S > (1 << 30))
            return MIFFT_E_UNSUPPORTED;
        *kind = 1;
        *split = inter_out ? 0 : 1;
        key[0] = (int)p0->S; key[1] = p0->L; key[2] = p1->L;
        return 0;
    }
    return MIFFT_E_UNSUPPORTED;
}

int pair_call(int precision, int kind, const int key[3], int split, const mifft::PairArgs* a, hipStream_t s, int query, int* width) {
    if (precision == MIFFT_F64) return mifft_pair_f64(kind, key[0], key[1], key[2], split, a, s, query, width);
    if (precision == MIFFT_F32) return mifft_pair_f32(kind, key[0], key[1], key[2], split, a, s, query, width);
    return MIFFT_E_UNSUPPORTED;
}

// Last-level cache size and XCD count of a HIP device: HIP has no attribute for either, the HSA agent behind the device has
// (matched by PCI domain / bus / device / function).  HSA is already initialised by the HIP runtime; hsa_init / hsa_shut_down only
// move its reference count.
struct HsaProbe {
    uint32_t domain, bdf;
    uint32_t cache[4];
    uint32_t xcc;
    bool found;
};
hsa_status_t hsa_probe_agent(hsa_agent_t agent, void* data) {
    HsaProbe* p = static_cast<HsaProbe*>(data);
    hsa_device_type_t type;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    if (hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    if (bdf != p->bdf || domain != p->domain) return HSA_STATUS_SUCCESS;
    uint32_t cache[4] = {0, 0, 0, 0}, xcc = 0;
    if (hsa_agent_get_info(agent, HSA_AGENT_INFO_CACHE_SIZE, cache) == HSA_STATUS_SUCCESS) memcpy(p->cache, cache, sizeof(cache));
    if (hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_NUM_XCC, &xcc) == HSA_STATUS_SUCCESS) p->xcc = xcc;
    p->found = true;
    return HSA_STATUS_INFO_BREAK;
}
void probe_memory_system(const hipDeviceProp_t& hp, int64_t* llc_bytes, int32_t* num_xcc) {
    HsaProbe probe;
    memset(&probe, 0, sizeof(probe));
    probe.domain = (uint32_t)hp.pciDomainID;
    probe.bdf = ((uint32_t)hp.pciBusID << 8) | ((uint32_t)hp.pciDeviceID << 3);
    if (hsa_init() == HSA_STATUS_SUCCESS) {
        (void)hsa_iterate_agents(hsa_probe_agent, &probe);
        (void)hsa_shut_down();
    }
    *llc_bytes = probe.found ? (int64_t)probe.cache[2] : 0;
    *num_xcc = (probe.found && probe.xcc >= 1) ? (int32_t)probe.xcc : 0;
    // a partition (CPX / DPX / QPX: an agent of 1, 2 or 4 of the 8 XCDs) may still report the whole part's Infinity Cache; the other
    // partitions stream through it too, so the planner gets this agent's FAIR SHARE (not verified on partitioned hardware: the sizes
    // derived from it are only ever too small, never too large)
    if (probe.found && probe.xcc >= 1 && probe.xcc < 8 && hp.multiProcessorCount < 228 && *llc_bytes >= (int64_t)(256ll << 20) &&
        (strncmp(hp.gcnArchName, "gfx94", 5) == 0 || strncmp(hp.gcnArchName, "gfx95", 5) == 0))
        *llc_bytes = *llc_bytes * (int64_t)probe.xcc / 8;
    if (*llc_bytes == 0 && (strncmp(hp.gcnArchName, "gfx94", 5) == 0 || strncmp(hp.gcnArchName, "gfx95", 5) == 0)) {
        // no answer from HSA on a CDNA3/4 part: the documented 256 MiB per 8 XCDs, scaled to the partition this device is
        *llc_bytes = (int64_t)(256ll << 20) * (hp.multiProcessorCount >= 228 ? 8 : (hp.multiProcessorCount + 37) / 38) / 8;
    }
    if (*num_xcc == 0) *num_xcc = hp.multiProcessorCount >= 228 ? 8 : (hp.multiProcessorCount >= 64 ? hp.multiProcessorCount / 32 : 1);
}

// which kernel runs dense smooth rows of n points by default (profiles/r04_j_mixed_rows_ab.log): fp64 the single-buffer tile kernel
// (N = 1000 0.503 -> 0.569, 2000 0.382 -> 0.460, 100 0.661 -> 0.685); fp32 the two-buffer row kernel (N = 1000 0.513 against 0.445)
// except for the longest rows, which fill its tile alone (N = 4000 0.279 -> 0.337; 3125 0.313 / 0.305, 3000 0.340 / 0.316)
bool mixed_rows_prefer_nd(bool f64, int n) {
    return f64 || n > 3200;
}

// work-list control block of a persistent launch from the caller's mifft_fused_sync (include/mifft.h): validates, zeroes the
// counters on `stream` when the caller does not alternate between two sets
int fill_ctl(mifft::FusedCtl* c, const mifft_fused_sync* sync, long long outer, int lag, int ring_slots, unsigned tiles0, unsigned tiles1,
             hipStream_t stream, const char* who) {
    static_assert(mifft::kFusedCS == MIFFT_FUSED2_COUNTER_STRIDE, "counter stride");
    if (!sync || !sync->counters) return set_err(MIFFT_E_INVALID, "%s: null counters", who);
    if (((uintptr_t)sync->counters | (uintptr_t)sync->counters_next) & 255) return set_err(MIFFT_E_INVALID, "%s: counter buffers must be 256-byte aligned", who);
    if ((uintptr_t)sync->error_word & 3) return set_err(MIFFT_E_INVALID, "%s: misaligned error word", who);
    if (sync->counters_next == sync->counters) return set_err(MIFFT_E_INVALID, "%s: counters_next must be a second buffer", who);
    if (outer > 0x3fffffff) return set_err(MIFFT_E_INVALID, "%s: batch too large", who);
    if (sync->counters_next) {
        // the launch zeroes words 0 and 1 of EVERY line of counters_next -- word 1 of line 0 is the default error word of the launch
        // that ran on that set, so a time-out report would vanish with the next launch
        if (!sync->error_word) return set_err(MIFFT_E_INVALID, "%s: alternating counter sets need an error word of their own", who);
        // a CAPTURED launch replays on the same set every time, while the two-set form relies on the host alternating the sets per
        // launch: the second replay would start on non-zero counters.  Captured launches take the single-set form, whose memset
        // becomes a node of the graph in front of the kernel
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive)
            return set_err(MIFFT_E_INVALID, "%s: a launch on a capturing stream must use the single-set form (counters_next = NULL)", who);
    }
    c->counters = (unsigned*)sync->counters;
    c->counters_next = (unsigned*)sync->counters_next;
    c->err = sync->error_word ? (unsigned*)sync->error_word : (unsigned*)sync->counters + 1;
    c->lines = (unsigned)(9 + 2 * outer);
    c->batch = (unsigned)outer;
    c->lag = (unsigned)lag;
    c->ring = (unsigned)ring_slots;
    c->tiles0 = tiles0;
    c->tiles1 = tiles1;
    if (!sync->counters_next) {
        // single-set form: zero the set in front of the launch.  On a CAPTURING stream with a kernel of our own (a kernel node) instead
        // of a memset node: see mifft_aux_zero_launch (csrc/fft_aux.hip)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive) {
            const int rz = mifft_aux_zero_launch(sync->counters, MIFFT_FUSED2_COUNTER_BYTES(outer), stream);
            return rz == 0 ? 0 : hip_check((hipError_t)rz, "kernel launch");
        }
        return hip_check(hipMemsetAsync(sync->counters, 0, MIFFT_FUSED2_COUNTER_BYTES(outer), stream), "hipMemsetAsync");
    }
    return 0;
}

}  // namespace

extern "C" {

int mifft_abi_version(void) { return MIFFT_ABI_VERSION; }
int mifft_debug_set(int32_t key, int32_t value) {
    if (key < 0 || key >= MIFFT_DEBUG_KEYS) return set_err(MIFFT_E_INVALID, "bad debug key %d", key);
    t_debug_value[key] = value;
    t_debug_mask |= 1u << key;
    return 0;
}
int mifft_debug_set_default(int32_t key, int32_t value) {
    if (key < 0 || key >= MIFFT_DEBUG_KEYS) return set_err(MIFFT_E_INVALID, "bad debug key %d", key);
    g_debug_default[key].store(value, std::memory_order_relaxed);
    return 0;
}
int mifft_has_feature(int32_t feature) {
#ifdef MIFFT_DEV_BUILD
    return (feature >= MIFFT_FEATURE_XCD2 && feature <= MIFFT_FEATURE_AB_FORMS) ? 1 : 0;
#else
    (void)feature;
    return 0;
#endif
}
int mifft_debug_get(int32_t key) { return (key >= 0 && key < MIFFT_DEBUG_KEYS) ? g_debug[key] : 0; }
const char* mifft_last_error(void) { return g_err; }

int mifft_device_count(int* count) {
    if (!count) return set_err(MIFFT_E_INVALID, "null argument");
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) {
        *count = 0;
        (void)hipGetLastError();
        return 0;
    }
    return hip_check(e, "hipGetDeviceCount");
}
int mifft_set_device(int device) { return hip_check(hipSetDevice(device), "hipSetDevice"); }
int mifft_get_device(int* device) { return hip_check(hipGetDevice(device), "hipGetDevice"); }

int mifft_device_props_get(int device, mifft_device_props* props) {
    if (!props) return set_err(MIFFT_E_INVALID, "null argument");
    hipDeviceProp_t p;
    int rc = hip_check(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
    if (rc) return rc;
    memset(props, 0, sizeof(*props));
    snprintf(props->name, sizeof(props->name), "%s", p.name);
    snprintf(props->gcn_arch, sizeof(props->gcn_arch), "%s", p.gcnArchName);
    props->compute_units = p.multiProcessorCount;
    props->wavefront_size = p.warpSize;
    props->max_threads_per_block = p.maxThreadsPerBlock;
    props->max_grid_x = p.maxGridSize[0];
    props->lds_bytes_per_block = (int64_t)p.sharedMemPerBlock;
    props->total_mem_bytes = (int64_t)p.totalGlobalMem;
    props->clock_khz = p.clockRate;
    props->l2_bytes = p.l2CacheSize;
    probe_memory_system(p, &props->llc_bytes, &props->num_xcc);
    return 0;
}

int mifft_malloc(void** ptr, size_t nbytes) {
    if (!ptr) return set_err(MIFFT_E_INVALID, "null argument");
    return hip_check(hipMalloc(ptr, nbytes ? nbytes : 1), "hipMalloc");
}
int mifft_free(void* ptr) { return hip_check(hipFree(ptr), "hipFree"); }
int mifft_memset(void* ptr, int value, size_t nbytes, mifft_stream_t stream) {
    return hip_check(hipMemsetAsync(ptr, value, nbytes, (hipStream_t)stream), "hipMemsetAsync");
}
int mifft_memcpy_h2d(void* dst, const void* src, size_t nbytes, mifft_stream_t stream) {
    int rc = hip_check(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync(h2d)");
    if (rc) return rc;
    return hip_check(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
}
int mifft_memcpy_d2h(void* dst, const void* src, size_t nbytes, mifft_stream_t stream) {
    int rc = hip_check(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync(d2h)");
    if (rc) return rc;
    return hip_check(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
}
int mifft_memcpy_d2d(void* dst, const void* src, size_t nbytes, mifft_stream_t stream) {
    return hip_check(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync(d2d)");
}
int mifft_stream_create(mifft_stream_t* stream) {
    if (!stream) return set_err(MIFFT_E_INVALID, "null argument");
    hipStream_t s;
    // a BLOCKING stream, like PyCUDA's (cuda.py:94-96): null-stream copies (DeviceArray.get/set, the reference quick-start
    // doc/source/index.rst:61-92) and a framework's default-stream kernels order against the plan's work without a sync
    int rc = hip_check(hipStreamCreateWithFlags(&s, hipStreamDefault), "hipStreamCreate");
    if (rc) return rc;
    *stream = (mifft_stream_t)s;
    return 0;
}
int mifft_host_alloc(void** ptr, size_t nbytes) {
    if (!ptr) return set_err(MIFFT_E_INVALID, "null argument");
    return hip_check(hipHostMalloc(ptr, nbytes ? nbytes : 1, hipHostMallocDefault), "hipHostMalloc");
}
int mifft_host_free(void* ptr) { return hip_check(hipHostFree(ptr), "hipHostFree"); }
int mifft_memcpy_d2h_async(void* dst, const void* src, size_t nbytes, mifft_stream_t stream) {
    return hip_check(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync(d2h)");
}
int mifft_event_query(mifft_event_t event) {
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 1;
    }
    return hip_check(e, "hipEventQuery");
}
int mifft_stream_wait_event(mifft_stream_t stream, mifft_event_t event) {
    return hip_check(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0), "hipStreamWaitEvent");
}
int mifft_stream_destroy(mifft_stream_t stream) { return hip_check(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy"); }
int mifft_stream_sync(mifft_stream_t stream) { return hip_check(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"); }
int mifft_stream_is_capturing(mifft_stream_t stream, int32_t* capturing) {
    if (!capturing) return set_err(MIFFT_E_INVALID, "null argument");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const int rc = hip_check(hipStreamIsCapturing((hipStream_t)stream, &st), "hipStreamIsCapturing");
    if (rc) return rc;
    *capturing = st == hipStreamCaptureStatusActive ? 1 : 0;
    return 0;
}
int mifft_stream_begin_capture(mifft_stream_t stream) {
    if (!stream) return set_err(MIFFT_E_INVALID, "the default stream cannot be captured");
    return hip_check(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
}
int mifft_stream_end_capture(mifft_stream_t stream, mifft_graph_t* graph) {
    if (!graph) return set_err(MIFFT_E_INVALID, "null argument");
    hipGraph_t g = nullptr;
    int rc = hip_check(hipStreamEndCapture((hipStream_t)stream, &g), "hipStreamEndCapture");
    if (rc) return rc;
    hipGraphExec_t ge = nullptr;
    rc = hip_check(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0), "hipGraphInstantiate");
    (void)hipGraphDestroy(g);
    if (rc) return rc;
    *graph = (mifft_graph_t)ge;
    return 0;
}
int mifft_graph_launch(mifft_graph_t graph, mifft_stream_t stream) {
    if (!graph) return set_err(MIFFT_E_INVALID, "null graph");
    return hip_check(hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream), "hipGraphLaunch");
}
int mifft_graph_destroy(mifft_graph_t graph) {
    if (!graph) return 0;
    return hip_check(hipGraphExecDestroy((hipGraphExec_t)graph), "hipGraphExecDestroy");
}
int mifft_device_sync(void) { return hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize"); }
int mifft_event_create(mifft_event_t* event) {
    if (!event) return set_err(MIFFT_E_INVALID, "null argument");
    hipEvent_t e;
    int rc = hip_check(hipEventCreate(&e), "hipEventCreate");
    if (rc) return rc;
    *event = (mifft_event_t)e;
    return 0;
}
int mifft_event_destroy(mifft_event_t event) { return hip_check(hipEventDestroy((hipEvent_t)event), "hipEventDestroy"); }
int mifft_event_record(mifft_event_t event, mifft_stream_t stream) {
    return hip_check(hipEventRecord((hipEvent_t)event, (hipStream_t)stream), "hipEventRecord");
}
int mifft_event_sync(mifft_event_t event) { return hip_check(hipEventSynchronize((hipEvent_t)event), "hipEventSynchronize"); }
int mifft_event_elapsed_ms(float* ms, mifft_event_t start, mifft_event_t stop) {
    return hip_check(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
}

int mifft_nd_max_points_for(int32_t precision) { return mifft_nd_max_points(precision == MIFFT_F64); }

int mifft_nd_shape_supported(int32_t precision, int32_t x, int32_t y, int32_t z, int32_t variant) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    if (x < 1 || y < 1 || z < 1 || (x & (x - 1)) || (y & (y - 1)) || (z & (z - 1))) return MIFFT_E_UNSUPPORTED;
    const long long n = (long long)x * y * z;
    if (n <= mifft_nd_max_points(precision == MIFFT_F64)) return 0;
    if (variant == MIFFT_VARIANT_SPLIT_ONLY)      // planes on both sides: the tiled fixed-shape kernel with one tile per parent (launch_nd)
        // (x rows of >= 256 bytes per plane: fp64 (128, 128) 0.374 as two passes -> 0.568, (16, 32, 32) 0.338 -> 0.499; fp32 32^3, 128-byte
        // rows, measured 0.285 against 0.307 for its two passes and keeps them -- profiles/r04_at_rows_split.log)
        // (round 6: or a dense planes instance of fft_nd2p.hip -- fp32 (16, 16, 128), 32^3)
        return (g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && g_debug[MIFFT_DEBUG_NO_ND2] == 0 &&
                ((x * (precision == MIFFT_F64 ? 8 : 4) >= 256 && mifft_nd2t_split(precision == MIFFT_F64, x, y, z, nullptr, nullptr, nullptr, 1) == 0) ||
                 (g_debug[MIFFT_DEBUG_ALT_ROWS] != 7 && mifft_nd2p(precision == MIFFT_F64 ? 1 : 0, x, y, z, nullptr, nullptr, 1) == 0))) ? 0 : MIFFT_E_UNSUPPORTED;
    if (variant == MIFFT_VARIANT_OUT_OF_PLACE_ONLY || variant == MIFFT_VARIANT_OUT_OF_PLACE_ANY_SIZE)
        // interleaved on both sides AND out of place: several work-groups per transform (fft_nd2z.hpp)
        return (g_debug[MIFFT_DEBUG_NO_ND2] == 0 && g_debug[MIFFT_DEBUG_ALT_ROWS] != 6 &&
                mifft_nd2z(precision == MIFFT_F64 ? 1 : 0, x, y, z, nullptr, nullptr, variant == MIFFT_VARIANT_OUT_OF_PLACE_ONLY ? 1 : 2) == 0)
                   ? 0 : MIFFT_E_UNSUPPORTED;
    if (variant == MIFFT_VARIANT_SPLIT_OUT_OF_PLACE || variant == MIFFT_VARIANT_SPLIT_OUT_OF_PLACE_ANY_SIZE)
        // planes on both sides AND out of place: several work-groups per transform on 16-byte plane accesses (fft_nd2zp.hpp)
        return (g_debug[MIFFT_DEBUG_NO_ND2] == 0 && g_debug[MIFFT_DEBUG_ALT_ROWS] != 6 && g_debug[MIFFT_DEBUG_ALT_ROWS] != 7 &&
                g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 &&
                mifft_nd2zp(precision == MIFFT_F64 ? 1 : 0, x, y, z, nullptr, nullptr, variant == MIFFT_VARIANT_SPLIT_OUT_OF_PLACE ? 1 : 2) == 0)
                   ? 0 : MIFFT_E_UNSUPPORTED;
    if (variant != MIFFT_VARIANT_INTERLEAVED_ONLY) return MIFFT_E_UNSUPPORTED;
    const int rc = precision == MIFFT_F64 ? mifft_nd2_f64_supported(x, y, z) : mifft_nd2_f32_supported(x, y, z);
    return rc == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_pass_supported(int32_t kind, int32_t precision, int32_t L, int32_t variant) {
    if (kind == MIFFT_PASS_ND) return (L >= 1 && L <= mifft_nd_max_points(precision == MIFFT_F64) && (L & (L - 1)) == 0) ? 0 : MIFFT_E_UNSUPPORTED;
    mifft_pass p;
    memset(&p, 0, sizeof(p));
    p.kind = kind;
    p.precision = precision;
    p.L = L;
    p.variant = variant;
    p.M = 1;
    p.S = (kind == MIFFT_PASS_COL) ? 2 : 1;
    int rc0 = dispatch(&p, nullptr, nullptr, 1);
    if (rc0 || kind != MIFFT_PASS_COL) return rc0;
    p.S = 1;  // transposing form
    p.M = 2;
    return dispatch(&p, nullptr, nullptr, 1);
}

int mifft_launch_pass(const mifft_pass* p, const void* in0, const void* in1, void* out0, void* out1, mifft_stream_t stream) {
    int rc = validate(p);
    if (rc) return rc;
    if (!in0 || !out0) return set_err(MIFFT_E_INVALID, "null data buffer");
    const bool split_in = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_SRC_INTERLEAVED);
    const bool split_out = p->layout == MIFFT_SPLIT && !(p->flags & MIFFT_FLAG_DST_INTERLEAVED);
    const bool split = split_in;
    if ((split_in && !in1) || (split_out && !out1)) return set_err(MIFFT_E_INVALID, "split layout needs imaginary planes");
    if (p->layout != MIFFT_SPLIT && (in1 || out1)) return set_err(MIFFT_E_INVALID, "interleaved layout takes no imaginary planes");
    if (!split_in) in1 = nullptr;
    if (!split_out) out1 = nullptr;
    const uintptr_t align_mask = 15;
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)in1 | (uintptr_t)out1) & align_mask)
        return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if ((p->outer_stride_in | p->outer_stride_out) & 1) return set_err(MIFFT_E_INVALID, "outer strides must be even");
    if (p->kind == MIFFT_PASS_COL && p->M > 1 && (in0 == out0 || (split && split_out && in1 == out1)))
        return set_err(MIFFT_E_INVALID, "a COL pass with M > 1 cannot run in place");
    if (p->outer == 0) return 0;

    if (p->kind == MIFFT_PASS_ND) {
        if ((p->outer * p->L * p->M * p->S) & 3) return set_err(MIFFT_E_INVALID, "ND pass: total points must be a multiple of 4");
        return launch_nd(p, in0, in1, out0, out1, (hipStream_t)stream);
    }
    mifft::TileArgs a;
    fill_args(p, in0, in1, out0, out1, &a);
    return dispatch(p, &a, (hipStream_t)stream, 0);
}


int mifft_pair_split(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z) {
    if (x < 2 || y < 2 || z < 2 || !is_pow2(x) || !is_pow2(y) || !is_pow2(z)) return 0;
    const int split = layout == MIFFT_SPLIT ? 1 : 0;
    // the development switch "no pass pairs" acts HERE, when a plan is built, and nowhere else: a plan that was built with pairs
    // keeps launching them whatever the switch says later
    if (g_debug[MIFFT_DEBUG_PAIR] == 1) return 0;
    // candidates in order of preference (measured: profiles/r03_b_c4_pair_split.log); MIFFT_DEBUG_PAIR = 2 swaps them
    int cand[2] = {32, 64};
    if (g_debug[MIFFT_DEBUG_PAIR] == 2) { cand[0] = 64; cand[1] = 32; }
    for (int r0 : cand) {
        if (y % r0 || y / r0 < 2) continue;
        const int kxy[3] = {x, r0, y / r0}, kyz[3] = {x * r0, y / r0, z};
        if (pair_call(precision, 0, kxy, split, nullptr, nullptr, 1, nullptr) == 0 &&
            pair_call(precision, 1, kyz, split, nullptr, nullptr, 1, nullptr) == 0)
            return r0;
    }
    return 0;
}

int mifft_pair_kernel_supported(int32_t precision, int32_t layout, int32_t kind, int32_t k0, int32_t k1, int32_t k2) {
    if ((precision != MIFFT_F32 && precision != MIFFT_F64) || (layout != MIFFT_INTERLEAVED && layout != MIFFT_SPLIT) || (kind != 0 && kind != 1))
        return MIFFT_E_UNSUPPORTED;
    if (g_debug[MIFFT_DEBUG_PAIR] == 1) return MIFFT_E_UNSUPPORTED;      // (as mifft_pair_split: the switch acts when a plan is built)
    const int key[3] = {k0, k1, k2};
    return pair_call(precision, kind, key, layout == MIFFT_SPLIT ? 1 : 0, nullptr, nullptr, 1, nullptr) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_pass_pair_supported(const mifft_pass* p0, const mifft_pass* p1) {
    int kind, key[3], split;
    int rc = classify_pair(p0, p1, &kind, key, &split);
    if (rc) return MIFFT_E_UNSUPPORTED;
    return pair_call(p0->precision, kind, key, split, nullptr, nullptr, 1, nullptr) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_launch_pass_pair(const mifft_pass* p0, const mifft_pass* p1, const void* in0, const void* in1, void* out0, void* out1,
                           mifft_stream_t stream) {
    int rc = validate(p0);
    if (rc) return rc;
    rc = validate(p1);
    if (rc) return rc;
    int kind, key[3], split;
    if (classify_pair(p0, p1, &kind, key, &split) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "pass pair: not a (ROW x, COL y) / (COL y, COL z) pair of a dense batch with an interleaved intermediate");
    if (!in0 || !out0) return set_err(MIFFT_E_INVALID, "null data buffer");
    if (split && ((kind == 0 && !in1) || (kind == 1 && !out1))) return set_err(MIFFT_E_INVALID, "split layout needs imaginary planes");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)in1 | (uintptr_t)out1) & 15) return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if (kind == 0 && in0 == out0) return set_err(MIFFT_E_INVALID, "the (ROW x, COL y) pair cannot run in place");
    if (p1->outer == 0) return 0;
    int width = 0;
    if (pair_call(p0->precision, kind, key, split, nullptr, nullptr, 1, &width) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "pass pair: no kernel for kind %d keys (%d, %d, %d) split %d", kind, key[0], key[1], key[2], split);
    mifft::PairArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0;
    a.in1 = (kind == 0 && split) ? in1 : nullptr;
    a.out0 = out0;
    a.out1 = (kind == 1 && split) ? out1 : nullptr;
    if (kind == 0) {
        a.tw[0] = p0->tw_L;   // w(nx)
        a.tw[1] = p1->tw_L;   // w(R0)
        a.tw_lo = p1->tw_lo;  // w(ny), two-level
        a.tw_hi = p1->tw_hi;
        a.tw_shift = p1->tw_shift;
        a.tiles = p1->outer * p1->M;                 // planes * R1
    } else {
        a.tw[1] = p0->tw_L;   // w(R1)
        a.tw[2] = p1->tw_L;   // w(nz)
        a.tiles = p1->outer * (p0->S / width);       // transforms * groups of `width` adjacent columns
    }
    a.inverse = p0->inverse ? 1 : 0;
    // streamed sides: non-temporal loads of the plan's input, PLAIN stores on both launches (BASELINE config 4 at batch 64, one
    // box: plain 13.31 ms, write-through 14.44 ms; non-temporal stores measured below plain ones at batch 16 --
    // profiles/r03_c_store_policy.log)
    a.nt = (p0->flags & MIFFT_FLAG_STREAM_SRC) ? 1 : 0;
    if (p1->flags & MIFFT_FLAG_WRITE_THROUGH) a.nt |= 4;
    if (g_debug[MIFFT_DEBUG_STORE] == 1) a.nt = (a.nt & 1) | 2;
    else if (g_debug[MIFFT_DEBUG_STORE] == 2) a.nt = (a.nt & 1) | 4;
    else if (g_debug[MIFFT_DEBUG_STORE] == 3) a.nt = a.nt & 1;
    a.scale = p0->scale * p1->scale;
    rc = pair_call(p0->precision, kind, key, split, &a, (hipStream_t)stream, 0, nullptr);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

// one unit of a chain: a single pass, or a pass pair (MIFFT_FLAG_PAIR_WITH_NEXT on the first); *consumed = descriptors used
static int launch_unit(const mifft_pass* passes, int32_t i, int32_t npasses, const void* in0, const void* in1, void* out0, void* out1,
                       mifft_stream_t stream, int* consumed) {
    const mifft_pass* p = &passes[i];
    if (p->flags & MIFFT_FLAG_PAIR_WITH_NEXT) {
        if (i + 1 >= npasses) return set_err(MIFFT_E_INVALID, "pass %d: MIFFT_FLAG_PAIR_WITH_NEXT on the last pass", i);
        *consumed = 2;
        return mifft_launch_pass_pair(p, &passes[i + 1], in0, in1, out0, out1, stream);
    }
    *consumed = 1;
    return mifft_launch_pass(p, in0, in1, out0, out1, stream);
}

int mifft_launch_chain(const mifft_pass* passes, int32_t npasses, void* const bufs0[3], void* const bufs1[3], mifft_stream_t stream) {
    if (npasses < 0 || (npasses > 0 && !passes) || !bufs0) return set_err(MIFFT_E_INVALID, "bad chain arguments");
    for (int i = 0; i < npasses; ++i) {
        const mifft_pass* p = &passes[i];
        if (p->src < 0 || p->src > 2 || p->dst < 0 || p->dst > 2) return set_err(MIFFT_E_INVALID, "pass %d: bad buffer index", i);
    }
    for (int i = 0; i < npasses;) {
        const mifft_pass* p = &passes[i];
        const mifft_pass* pl = (p->flags & MIFFT_FLAG_PAIR_WITH_NEXT) && i + 1 < npasses ? &passes[i + 1] : p;   // the unit writes pl->dst
        const void* i1 = bufs1 ? bufs1[p->src] : nullptr;
        void* o1 = bufs1 ? bufs1[pl->dst] : nullptr;
        int used = 1;
        int rc = launch_unit(passes, i, npasses, bufs0[p->src], i1, bufs0[pl->dst], o1, stream, &used);
        if (rc) return rc;
        i += used;
    }
    return 0;
}

int mifft_launch_fused2(const mifft_pass* p0, const mifft_pass* p1, const void* in0, const void* in1, void* out0, void* out1,
                        void* ring0, void* ring1, int32_t ring_slots, int32_t lag, const mifft_fused_sync* sync, int32_t grid,
                        mifft_stream_t stream) {
    int rc = validate(p0);
    if (rc) return rc;
    rc = validate(p1);
    if (rc) return rc;
    if (p0->precision != p1->precision) return set_err(MIFFT_E_INVALID, "fused2: passes of two precisions");
    const bool f64 = p0->precision == MIFFT_F64;
    // 2-D form: the ROW pass (x axis, length nx) and the strided COL pass (y axis, length ny) of a 2-D plan, run as two
    // transposing column passes; fp32: nx, ny in {512, 1024, 2048} (split planes: squares only), fp64: 1024 x 1024
    const bool twod = p0->kind == MIFFT_PASS_ROW;
    if (twod) {
        auto side = [](int L) { return L == 512 || L == 1024 || L == 2048; };
        auto side64 = [](int L) { return L == 512 || L == 1024; };
        // a 256-point axis (interleaved): fp32 on the 32-column tiles next to a side <= 1024, fp64 next to a side <= 512
        const bool small = p0->layout != MIFFT_SPLIT && (p0->L == 256 || p1->L == 256) &&
                           (f64 ? (p0->L <= 512 && p1->L <= 512 && p0->L >= 256 && p1->L >= 256)
                                : (g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && mifft_fused2dw_f32(p1->L, p0->L, nullptr, 0, nullptr, 1, nullptr, nullptr) == 0));
        // split-complex fp32: the row-first kernel (fft_fused2r.hpp), (ny, nx) in {256, 512, 1024}^2
        const bool rowfirst_ok = !f64 && p0->layout == MIFFT_SPLIT && g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && !g_debug[MIFFT_DEBUG_NO_ROWFIRST] &&
                                 mifft_fused2r_f32(p1->L, p0->L, nullptr, 0, nullptr, 1) == 0;
        const bool okL = small || rowfirst_ok || (f64 ? (side64(p0->L) && side64(p1->L) && ((p0->L == 1024 && p1->L == 1024) || p0->layout != MIFFT_SPLIT))
                             : (side(p0->L) && side(p1->L) && (p0->L == p1->L || p0->layout != MIFFT_SPLIT)));
        if (p1->kind != MIFFT_PASS_COL || !okL || p1->S != p0->L || p1->M != 1 ||
            p0->outer != p1->outer * p1->L || p0->layout != p1->layout || p0->inverse != p1->inverse)
            return set_err(MIFFT_E_UNSUPPORTED, "fused2: the 2-D form takes (ny, nx) in {512, 1024, 2048}^2 (fp32; interleaved: also a 256-point side next to one <= 1024; split planes: {256, 512, 1024}^2 and the 2048 square) / {256, 512, 1024}^2 (fp64; split planes: 1024 x 1024; 256 only next to <= 512)");
    } else {
    if (p0->kind != MIFFT_PASS_COL || p1->kind != MIFFT_PASS_COL || p0->S != 1 || p0->M != p1->L || p1->M != 1 ||
        p1->S != p0->L || p0->outer != p1->outer || p0->layout != p1->layout || p0->inverse != p1->inverse)
        return set_err(MIFFT_E_INVALID, "fused2: passes are not the two passes of one long contiguous axis");
    auto ok_len = [](int L) { return L == 256 || L == 512 || L == 1024; };
    const bool big = p0->L == 2048 && (p1->L == 2048 || p1->L == 1024);   // 512-thread tiles (fft_col3.hpp)
    const bool wide64 = f64 && p0->L == 2048 && (p1->L == 2048 || p1->L == 1024) && p0->layout == MIFFT_INTERLEAVED;   // fft_fusedx_f64.hip
    const bool small64 = f64 && (p0->L == 256 || p0->L == 512) && (p1->L == 256 || p1->L == 512) && p0->L >= p1->L;
    const bool mid64 = f64 && p0->L == 1024 && (p1->L == 1024 || p1->L == 512);
    if (f64 ? (!wide64 && !small64 && !mid64) : (!big && (!ok_len(p0->L) || !ok_len(p1->L)))) return set_err(MIFFT_E_UNSUPPORTED, "fused2: no kernel for %d x %d", p0->L, p1->L);
    }
    const bool wide64 = !twod && f64 && p0->L == 2048;
    const bool split = p0->layout == MIFFT_SPLIT;
    if (!in0 || !out0 || !ring0 || (split && (!in1 || !out1))) return set_err(MIFFT_E_INVALID, "fused2: null buffer");
    (void)ring1;  // the ring is always interleaved
    if (grid < 1) return set_err(MIFFT_E_INVALID, "fused2: grid >= 1");
    // lag == 0: the sequential list of a tiny batch, one ring slot per transform
    if (lag == 0 && !mifft_has_feature(MIFFT_FEATURE_SEQUENTIAL_LIST))
        return set_err(MIFFT_E_UNSUPPORTED, "fused2: the sequential list (lag == 0) is a development form, not in this build (make DEV=1)");
    if (lag == 0 ? ring_slots != p1->outer && p1->outer > 0 : (ring_slots < 2 || lag < 1 || lag >= ring_slots))
        return set_err(MIFFT_E_INVALID, "fused2: need 1 <= lag < ring_slots, or lag == 0 with ring_slots == outer");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)ring0 | (uintptr_t)in1 | (uintptr_t)out1) & 15)
        return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if (p1->outer == 0) return 0;
    const int64_t n = (int64_t)p0->L * p1->L;
    mifft::FusedArgs f;
    fill_args(p0, in0, in1, ring0, nullptr, &f.p0);
    fill_args(p1, ring0, nullptr, out0, out1, &f.p1);
    f.p0.ostride_out = n;  // ring slot pitch
    f.p1.ostride_in = n;
    // split-complex fp32 2-D: the chain's own order -- ROW x from the planes, COL y to the planes -- on the persistent list
    const bool rowfirst = twod && !f64 && p0->layout == MIFFT_SPLIT && g_debug[MIFFT_DEBUG_NARROW_TILES] != 1 && !g_debug[MIFFT_DEBUG_NO_ROWFIRST] &&
                          mifft_fused2r_f32(p1->L, p0->L, nullptr, 0, nullptr, 1) == 0;
    if (rowfirst) {
        f.p0.nt = 4;                // the ring is written through (the consumers acquire it, fft_fused2.hpp)
        f.p0.scale = p0->scale;
        f.p1.logMS = f.p1.logS = ilog2(p0->L);   // M = 1, S = nx
        f.p1.has_tw = 0;
    } else if (twod) {
        // pass 0 = the y axis as a transposing column pass over in[y][x] (ny rows of M = nx columns, S = 1) -> ring[x][ky], with the
        // COL pass's table w(ny); pass 1 = the x axis as the same pass over ring[x][ky] (nx rows of ny columns) -> out[ky][kx],
        // with the ROW pass's table w(nx)
        f.p0.tw_L = p1->tw_L;
        f.p1.tw_L = p0->tw_L;
        f.p0.ostride_in = p1->outer_stride_in;
        f.p0.logMS = ilog2(p0->L); f.p0.logS = 0; f.p0.has_tw = 0; f.p0.total = p1->outer * p0->L;
        f.p1.logMS = ilog2(p1->L); f.p1.logS = 0; f.p1.has_tw = 0; f.p1.total = p1->outer * p1->L;
    }
    // tiles per transform: 2-D: nx / 16 column tiles in pass 0, ny / 16 in pass 1; 1-D: L1 / 16 and L0 / 16 (fp64 2048-point
    // passes: the tile widths of fft_fusedx_f64.hip)
    unsigned tiles0 = twod ? (unsigned)(p0->L / 16) : (unsigned)(p0->M / 16), tiles1 = twod ? (unsigned)(p1->L / 16) : (unsigned)(p1->S / 16);
    if (rowfirst) {                 // groups of 16 rows, then tiles of 16 columns
        tiles0 = (unsigned)(p1->L / 16);
        tiles1 = (unsigned)(p0->L / 16);
    }
    if (wide64 && mifft_fusedx_f64(p0->L, p1->L, nullptr, 0, nullptr, 1, &tiles0, &tiles1) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "fused2: no kernel for %d x %d", p0->L, p1->L);
    // fp32 interleaved 2^16 ... 2^18: 32-column tiles (fft_col2w.hpp) -- half as many tiles per pass
    const bool narrow = g_debug[MIFFT_DEBUG_NARROW_TILES] == 1;
    const bool mixed32 = !twod && !f64 && !split && p0->L == 1024 && p1->L == 512 && !narrow;        // 2^19: the second pass only
    const bool wide32 = mixed32 || (!twod && !f64 && !split && p0->L <= 512 && p1->L <= 512 && p0->L >= p1->L && !narrow);
    if (wide32) {
        if (!mixed32) tiles0 /= 2;
        tiles1 /= 2;
    }
    // split-complex fp32 on the 256-thread tiles (1-D: L0, L1 <= 1024; 2-D: the 512 and 1024 squares): an item is the two sibling
    // 16-column tiles side by side in one 512-thread work-group (fft_fused2s_kernel)
    const bool siblings = !f64 && split && !narrow && !rowfirst && p0->L <= 1024 && p1->L <= 1024;
    if (siblings) {
        tiles0 /= 2;
        tiles1 /= 2;
    }
    // 2-D fp32 interleaved with a 512-point axis: that axis' pass on 32-column tiles (p1->L = ny, p0->L = nx)
    const bool wide2d = twod && !f64 && !split && !narrow && mifft_fused2dw_f32(p1->L, p0->L, nullptr, 0, nullptr, 1, &tiles0, &tiles1) == 0;
    rc = fill_ctl(&f.c, sync, p1->outer, lag, ring_slots, tiles0, tiles1, (hipStream_t)stream, "fused2");
    if (rc) return rc;
    if (lag == 0) {
        const bool wide = (f64 && (p0->L > 512 || p1->L > 512)) || p0->L == 2048 || p1->L == 2048 || siblings;       // 512- / 1024-thread tiles: one work-group per CU
        grid = resident_grid(grid, wide ? 1 : 2);
    }
    rc = rowfirst ? mifft_fused2r_f32(p1->L, p0->L, &f, (unsigned)grid, (hipStream_t)stream, 0) :
         wide2d ? mifft_fused2dw_f32(p1->L, p0->L, &f, (unsigned)grid, (hipStream_t)stream, 0, nullptr, nullptr) :
         wide32 ? mifft_fused2w_f32_launch(p0->L, p1->L, &f, (unsigned)grid, (hipStream_t)stream) :
         wide64 ? mifft_fusedx_f64(p0->L, p1->L, &f, (unsigned)grid, (hipStream_t)stream, 0, nullptr, nullptr) : twod ? (f64 ? mifft_fused3d_f64_launch(p1->L, p0->L, &f, split ? 1 : 0, (unsigned)grid, (hipStream_t)stream)
                     : mifft_fused2d_f32_launch(p1->L, p0->L, &f, split ? 1 : 0, (unsigned)grid, (hipStream_t)stream))
       : f64 ? mifft_fused3_f64_launch(p0->L, p1->L, &f, split ? 1 : 0, (unsigned)grid, (hipStream_t)stream)
             : mifft_fused2_f32_launch(p0->L, p1->L, &f, split ? 1 : 0, (unsigned)grid, (hipStream_t)stream);
    if (rc == MIFFT_E_UNSUPPORTED) return set_err(rc, "fused2: no kernel for %d x %d", p0->L, p1->L);
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_launch_fused2x(const mifft_pass* p0, const mifft_pass* p1, const void* in0, const void* in1, void* out0, void* out1, void* ring0,
                         int32_t ring_slots, int32_t lag, const mifft_fused_sync* sync, int32_t grid, mifft_stream_t stream) {
    if (!mifft_has_feature(MIFFT_FEATURE_FUSED2X))
        return set_err(MIFFT_E_UNSUPPORTED, "fused2x: a development strategy, not in this build of the library (make DEV=1; mifft_has_feature)");
    int rc = validate(p0);
    if (rc) return rc;
    rc = validate(p1);
    if (rc) return rc;
    if (p0->precision != MIFFT_F32 || p1->precision != MIFFT_F32 || p0->layout != p1->layout)
        return set_err(MIFFT_E_UNSUPPORTED, "fused2x: fp32 only, one layout on both sides");
    const bool split = p0->layout == MIFFT_SPLIT;
    if (p0->kind != MIFFT_PASS_COL || p1->kind != MIFFT_PASS_COL || p0->S != 1 || p0->M != p1->L || p1->M != 1 ||
        p1->S != p0->L || p0->outer != p1->outer || p0->inverse != p1->inverse)
        return set_err(MIFFT_E_INVALID, "fused2x: passes are not the two passes of one long contiguous axis");
    if (!in0 || !out0 || !ring0 || (split && (!in1 || !out1))) return set_err(MIFFT_E_INVALID, "fused2x: null buffer");
    if (ring_slots < 2 || lag < 1 || lag >= ring_slots || grid < 1) return set_err(MIFFT_E_INVALID, "fused2x: need 1 <= lag < ring_slots, grid >= 1");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)ring0 | (uintptr_t)in1 | (uintptr_t)out1) & 15)
        return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if (p1->outer == 0) return 0;
    const int64_t n = (int64_t)p0->L * p1->L;
    mifft::FusedArgs f;
    fill_args(p0, in0, in1, ring0, nullptr, &f.p0);
    fill_args(p1, ring0, nullptr, out0, out1, &f.p1);
    f.p0.ostride_out = n;
    f.p1.ostride_in = n;
    f.p0.split_out = 0;      // the ring is interleaved for both layouts
    f.p1.split = 0;
    rc = fill_ctl(&f.c, sync, p1->outer, lag, ring_slots, (unsigned)(p0->M / 16), (unsigned)(p1->S / 16), (hipStream_t)stream, "fused2x");
    if (rc) return rc;
    rc = mifft_fused2x_f32_launch(p0->L, p1->L, &f, split ? 1 : 0, (unsigned)grid, (hipStream_t)stream);
    if (rc == MIFFT_E_UNSUPPORTED) return set_err(rc, "fused2x: no kernel for %d x %d", p0->L, p1->L);
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_fused_pair_supported(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z) {
    if ((precision != MIFFT_F32 && precision != MIFFT_F64) || (layout != MIFFT_INTERLEAVED && layout != MIFFT_SPLIT)) return MIFFT_E_UNSUPPORTED;
    return mifft_fusedp(precision == MIFFT_F64, layout == MIFFT_SPLIT, x, y, z, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_fused_pair_split(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z) {
    if ((precision != MIFFT_F32 && precision != MIFFT_F64) || (layout != MIFFT_INTERLEAVED && layout != MIFFT_SPLIT)) return 0;
    int r0 = 0;
    if (mifft_fusedp(precision == MIFFT_F64, layout == MIFFT_SPLIT, x, y, z, nullptr, 0, nullptr, 1, &r0, nullptr, nullptr) != 0) return 0;
    return r0;
}

int mifft_launch_fused_pair(const mifft_pass* passes, const void* in0, const void* in1, void* out0, void* out1, void* ring0, int32_t ring_slots,
                            int32_t lag, const mifft_fused_sync* sync, int32_t grid, mifft_stream_t stream) {
    if (!passes) return set_err(MIFFT_E_INVALID, "fused pair: null pass list");
    for (int i = 0; i < 4; ++i) {
        const int rc = validate(&passes[i]);
        if (rc) return rc;
    }
    const mifft_pass *px = &passes[0], *py0 = &passes[1], *py1 = &passes[2], *pz = &passes[3];
    int kxy, kyz, keyxy[3], keyyz[3], sxy, syz;
    // split-complex plans: the XY pair reads the two planes, the YZ pair writes them, the ring between them is interleaved
    const bool split = px->layout == MIFFT_SPLIT;
    if (classify_pair(px, py0, &kxy, keyxy, &sxy) != 0 || classify_pair(py1, pz, &kyz, keyyz, &syz) != 0 || kxy != 0 || kyz != 1 ||
        sxy != (split ? 1 : 0) || syz != (split ? 1 : 0) || pz->layout != px->layout ||
        py1->S != (int64_t)px->L * py0->L || py1->L != py0->M || pz->outer != py0->outer / pz->L || py1->inverse != px->inverse)
        return set_err(MIFFT_E_INVALID, "fused pair: not the (ROW x, COL y R0) + (COL y R1, COL z) pairs of one dense 3-D batch with an interleaved buffer between them");
    const int nx = px->L, ny = py0->L * (int)py0->M, nz = pz->L;
    const bool f64 = px->precision == MIFFT_F64;
    int r0 = 0;
    unsigned tiles0 = 0, tiles1 = 0;
    if (mifft_fusedp(f64, split ? 1 : 0, nx, ny, nz, nullptr, 0, nullptr, 1, &r0, &tiles0, &tiles1) != 0 || r0 != py0->L)
        return set_err(MIFFT_E_UNSUPPORTED, "fused pair: no kernel for %d x %d x %d with y = %d x %lld%s", nz, ny, nx, py0->L, (long long)py0->M, split ? " (split planes)" : "");
    if (!in0 || !out0 || !ring0 || (split && (!in1 || !out1))) return set_err(MIFFT_E_INVALID, "fused pair: null buffer");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)ring0 | (uintptr_t)in1 | (uintptr_t)out1) & 15) return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if (ring0 == in0 || ring0 == out0) return set_err(MIFFT_E_INVALID, "fused pair: the ring must be a buffer of its own");
    if (grid < 1) return set_err(MIFFT_E_INVALID, "fused pair: grid >= 1");
    const long long batch = pz->outer;
    if (lag == 0 && !mifft_has_feature(MIFFT_FEATURE_SEQUENTIAL_LIST))
        return set_err(MIFFT_E_UNSUPPORTED, "fused pair: the sequential list (lag == 0) is a development form, not in this build (make DEV=1)");
    if (lag == 0 ? ring_slots != batch && batch > 0 : (ring_slots < 2 || lag < 1 || lag >= ring_slots))
        return set_err(MIFFT_E_INVALID, "fused pair: need 1 <= lag < ring_slots, or lag == 0 with ring_slots == outer");
    if (batch == 0) return 0;
    mifft::FusedPairArgs f;
    memset(&f, 0, sizeof(f));
    f.n = (long long)nx * ny * nz;
    f.a0.in0 = in0;
    f.a0.in1 = split ? in1 : nullptr;
    f.a0.out0 = ring0;
    f.a0.tw[0] = px->tw_L;     // w(nx)
    f.a0.tw[1] = py0->tw_L;    // w(R0)
    f.a0.tw_lo = py0->tw_lo;   // w(ny), two-level
    f.a0.tw_hi = py0->tw_hi;
    f.a0.tw_shift = py0->tw_shift;
    f.a0.inverse = px->inverse ? 1 : 0;
    f.a0.scale = px->scale * py0->scale;
    f.a1.in0 = ring0;
    f.a1.out0 = out0;
    f.a1.out1 = split ? out1 : nullptr;
    f.a1.tw[1] = py1->tw_L;    // w(R1)
    f.a1.tw[2] = pz->tw_L;     // w(nz)
    f.a1.inverse = f.a0.inverse;
    f.a1.scale = py1->scale * pz->scale;
    int rc = fill_ctl(&f.c, sync, batch, lag, ring_slots, tiles0, tiles1, (hipStream_t)stream, "fused pair");
    if (rc) return rc;
    if (lag == 0) grid = resident_grid(grid, 2);
    rc = mifft_fusedp(f64, split ? 1 : 0, nx, ny, nz, &f, (unsigned)grid, (hipStream_t)stream, 0, nullptr, nullptr, nullptr);
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_launch_xcd2(const mifft_pass* p0, const mifft_pass* p1, const void* in0, const void* in1, void* out0, void* out1,
                      void* scratch, void* control, int32_t flags, mifft_stream_t stream) {
#ifndef MIFFT_DEV_BUILD
    (void)p0; (void)p1; (void)in0; (void)in1; (void)out0; (void)out1; (void)scratch; (void)control; (void)flags; (void)stream;
    return set_err(MIFFT_E_UNSUPPORTED, "xcd2: a development strategy, not in this build of the library (make DEV=1; mifft_has_feature)");
#else
    int rc = validate(p0);
    if (rc) return rc;
    rc = validate(p1);
    if (rc) return rc;
    if (p0->precision != MIFFT_F32 || p1->precision != MIFFT_F32) return set_err(MIFFT_E_UNSUPPORTED, "xcd2: fp32 only");
    if (p0->kind != MIFFT_PASS_COL || p1->kind != MIFFT_PASS_COL || p0->S != 1 || p0->M != p1->L || p1->M != 1 ||
        p1->S != p0->L || p0->outer != p1->outer || p0->layout != p1->layout || p0->inverse != p1->inverse)
        return set_err(MIFFT_E_INVALID, "xcd2: passes are not the two passes of one long contiguous axis");
    if (p0->L != 1024 || p1->L != 1024) return set_err(MIFFT_E_UNSUPPORTED, "xcd2: no kernel for %d x %d", p0->L, p1->L);
    const bool split = p0->layout == MIFFT_SPLIT;
    if ((p0->flags & MIFFT_FLAG_SRC_INTERLEAVED) || (p1->flags & MIFFT_FLAG_DST_INTERLEAVED))
        return set_err(MIFFT_E_INVALID, "xcd2: the user sides follow the plan layout");
    if (!in0 || !out0 || !scratch || !control || (split && (!in1 || !out1))) return set_err(MIFFT_E_INVALID, "xcd2: null buffer");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)in1 | (uintptr_t)out1) & 15)
        return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if ((uintptr_t)scratch & 255) return set_err(MIFFT_E_INVALID, "xcd2: scratch must be 256-byte aligned");
    // in place is fine: every point of a transform is in registers before any of its output is stored
    if (p0->outer == 0) return 0;
    if (p0->outer > 0x0fffffff) return set_err(MIFFT_E_INVALID, "xcd2: batch too large");
    int dev = 0;
    rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    int cus = 0;
    rc = hip_check(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute");
    if (rc) return rc;
    if (cus != 256) return set_err(MIFFT_E_UNSUPPORTED, "xcd2: needs 8 XCDs x 32 CUs (device has %d CUs)", cus);
    mifft::Xcd2Args f;
    fill_args(p0, in0, in1, nullptr, nullptr, &f.p0);
    fill_args(p1, nullptr, nullptr, out0, out1, &f.p1);
    f.ctl = (unsigned*)control;
    f.scratch = scratch;
    f.batch = (unsigned)p0->outer;
    // development trace (MIFFT_XCD2_TRACE): 32 time stamps per work-group behind the control words
    f.trace = (flags & MIFFT_XCD2_TRACE) ? (unsigned long long*)((char*)control + MIFFT_XCD2_CONTROL_BYTES) : nullptr;
    f.trace_iter = (unsigned)((flags >> 8) & 0xffff);   // (flags bits 4..6: elimination mode, development)
    f.pace = (flags & 4) ? 1u : 0u;
    f.phase_us = (unsigned)((flags >> 24) & 0x7f);       // (flags bits 24..30: start offset of the odd XCDs in microseconds, development)
    rc = hip_check(hipMemsetAsync(control, 0, MIFFT_XCD2_CONTROL_BYTES, (hipStream_t)stream), "hipMemsetAsync");
    if (rc) return rc;
    rc = mifft_xcd2_f32_launch(&f, split ? 1 : 0, (flags & MIFFT_XCD2_PREFETCH) ? 1 : 0, (flags >> 4) & 7, 2u * (unsigned)cus, (hipStream_t)stream);
    if (rc == MIFFT_E_UNSUPPORTED) return set_err(rc, "xcd2: elimination modes exist for the interleaved prefetching form only");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
#endif
}

int mifft_launch_chain_pipelined(const mifft_pass* passes, int32_t npasses, void* const bufs0[3], void* const bufs1[3],
                                 int64_t batch, int64_t chunk, int64_t item_elems, mifft_stream_t stream,
                                 const mifft_stream_t* side, int32_t nside, const mifft_event_t* events) {
    if (npasses < 0 || (npasses > 0 && !passes) || !bufs0) return set_err(MIFFT_E_INVALID, "bad chain arguments");
    if (batch < 1 || chunk < 1 || item_elems < 1 || nside < 1 || !side || !events)
        return set_err(MIFFT_E_INVALID, "bad pipeline arguments");
    for (int i = 0; i < npasses; ++i) {
        const mifft_pass* p = &passes[i];
        if (p->src < 0 || p->src > 2 || p->dst < 0 || p->dst > 2) return set_err(MIFFT_E_INVALID, "pass %d: bad buffer index", i);
        if (p->outer % batch) return set_err(MIFFT_E_INVALID, "pass %d: outer is not a multiple of the batch", i);
    }
    if (npasses == 0) return 0;
    const bool f64 = passes[0].precision == MIFFT_F64;
    const bool split = passes[0].layout == MIFFT_SPLIT;
    const int64_t ebytes = (split ? 1 : 2) * (f64 ? 8 : 4);
    bool tmp_interleaved = !split;  // the temp buffer is interleaved when the passes touching it say so
    for (int i = 0; i < npasses; ++i) {
        if (passes[i].src == 2 && (passes[i].flags & MIFFT_FLAG_SRC_INTERLEAVED)) tmp_interleaved = true;
        if (passes[i].dst == 2 && (passes[i].flags & MIFFT_FLAG_DST_INTERLEAVED)) tmp_interleaved = true;
    }
    const int64_t tbytes = (tmp_interleaved ? 2 : 1) * (f64 ? 8 : 4);
    // nside == 1 with side[0] == stream: the chunks in order on the caller's own stream, no fork and no join (what a plan asks for when
    // its stream is being captured into a graph: a linear graph)
    const bool inline_chunks = nside == 1 && side[0] == stream;
    int rc = 0;
    const int64_t nchunks = (batch + chunk - 1) / chunk;
    const int used = (int)(nchunks < nside ? nchunks : nside);
    if (!inline_chunks) {
        rc = hip_check(hipEventRecord((hipEvent_t)events[0], (hipStream_t)stream), "hipEventRecord");
        if (rc) return rc;
        for (int s = 0; s < used; ++s) {
            rc = hip_check(hipStreamWaitEvent((hipStream_t)side[s], (hipEvent_t)events[0], 0), "hipStreamWaitEvent");
            if (rc) return rc;
        }
    }
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t nb = (c + 1) * chunk <= batch ? chunk : batch - c * chunk;
        const int slot = (int)(c % nside);
        const int64_t off_io = c * chunk * item_elems * ebytes;
        const int64_t off_tmp = (int64_t)slot * chunk * item_elems * tbytes;
        for (int i = 0; i < npasses;) {
            mifft_pass p[2] = {passes[i], passes[i + 1 < npasses ? i + 1 : i]};
            p[0].outer = p[0].outer / batch * nb;
            p[1].outer = p[1].outer / batch * nb;
            const bool pair = (p[0].flags & MIFFT_FLAG_PAIR_WITH_NEXT) && i + 1 < npasses;
            const int dst = pair ? p[1].dst : p[0].dst;
            auto at = [&](void* const* bufs, int idx) -> void* {
                if (!bufs || !bufs[idx]) return nullptr;
                return (char*)bufs[idx] + (idx == 2 ? off_tmp : off_io);
            };
            int used = 1;
            rc = launch_unit(p, 0, pair ? 2 : 1, at(bufs0, p[0].src), split ? at(bufs1, p[0].src) : nullptr, at(bufs0, dst),
                             split ? at(bufs1, dst) : nullptr, side[slot], &used);
            if (rc) return rc;
            i += used;
        }
    }
    for (int s = 0; s < used && !inline_chunks; ++s) {
        rc = hip_check(hipEventRecord((hipEvent_t)events[1 + s], (hipStream_t)side[s]), "hipEventRecord");
        if (rc) return rc;
        rc = hip_check(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)events[1 + s], 0), "hipStreamWaitEvent");
        if (rc) return rc;
    }
    return 0;
}

int mifft_nd_tiled_supported(int32_t precision, int32_t x, int32_t y, int32_t z) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    return mifft_nd2t(precision == MIFFT_F64, x, y, z, nullptr, nullptr, nullptr, 1) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

static int launch_nd_tiled(const mifft_pass* p, const mifft_tiling* t, const void* in0, const void* in1, void* out0, void* out1, bool split,
                           mifft_stream_t stream) {
    if (!p || !t) return set_err(MIFFT_E_INVALID, "nd_tiled: null argument");
    if (p->kind != MIFFT_PASS_ND) return set_err(MIFFT_E_INVALID, "nd_tiled: not an ND pass");
    if (p->precision != MIFFT_F32 && p->precision != MIFFT_F64) return set_err(MIFFT_E_INVALID, "bad precision %d", p->precision);
    if (p->layout != (split ? MIFFT_SPLIT : MIFFT_INTERLEAVED))
        return set_err(split ? MIFFT_E_INVALID : MIFFT_E_UNSUPPORTED, split ? "nd_tiled_split: the pass must say MIFFT_SPLIT" : "nd_tiled: interleaved data only (split planes: mifft_launch_nd_tiled_split)");
    if (mifft_nd_tiled_supported(p->precision, p->L, (int32_t)p->M, (int32_t)p->S) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "nd_tiled: no kernel for tiles of %d x %lld x %lld", p->L, (long long)p->M, (long long)p->S);
    if (t->cx < 1 || t->cy < 1 || t->cz < 1 || t->pitch_y < (int64_t)p->L * t->cx || t->pitch_z < t->pitch_y * p->M * t->cy ||
        t->parent_elems < t->pitch_z * p->S * t->cz || (t->pitch_y & 1) || (t->pitch_z & 1) || (t->parent_elems & 1))
        return set_err(MIFFT_E_INVALID, "nd_tiled: inconsistent tiling");
    if (!in0 || !out0 || (split && (!in1 || !out1))) return set_err(MIFFT_E_INVALID, "null data buffer");
    if (((uintptr_t)in0 | (uintptr_t)out0 | (uintptr_t)in1 | (uintptr_t)out1) & 15) return set_err(MIFFT_E_INVALID, "data buffers must be 16-byte aligned");
    if ((p->L > 1 && !p->tw_L) || (p->M > 1 && !p->tw_lo) || (p->S > 1 && !p->tw_hi)) return set_err(MIFFT_E_INVALID, "ND pass: twiddle table missing");
    if (p->outer < 0) return set_err(MIFFT_E_INVALID, "negative outer count");
    if (p->outer == 0) return 0;
    mifft::TileArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1;
    a.split = a.split_out = split ? 1 : 0;
    a.tw_L = p->tw_L; a.tw_lo = p->tw_lo; a.tw_hi = p->tw_hi;
    a.inverse = p->inverse ? 1 : 0;
    a.scale = p->scale;
    mifft::TiledGeom g;
    g.pitch_y = t->pitch_y; g.pitch_z = t->pitch_z; g.parent = t->parent_elems; g.tiles = p->outer;
    g.cx = t->cx; g.cy = t->cy; g.cz = t->cz;
    const int rc = (split ? mifft_nd2t_split : mifft_nd2t)(p->precision == MIFFT_F64, p->L, (int)p->M, (int)p->S, &a, &g, (hipStream_t)stream, 0);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_launch_nd_tiled(const mifft_pass* p, const mifft_tiling* t, const void* in, void* out, mifft_stream_t stream) {
    return launch_nd_tiled(p, t, in, nullptr, out, nullptr, false, stream);
}

int mifft_launch_nd_tiled_split(const mifft_pass* p, const mifft_tiling* t, const void* in_re, const void* in_im, void* out_re, void* out_im,
                                mifft_stream_t stream) {
    return launch_nd_tiled(p, t, in_re, in_im, out_re, out_im, true, stream);
}

int mifft_aux_copy(const mifft_copy* c, const void* src0, const void* src1, void* dst0, void* dst1, mifft_stream_t stream) {
    if (!c || c->ndim < 1 || c->ndim > 6) return set_err(MIFFT_E_INVALID, "aux_copy: bad descriptor");
    if (c->precision != MIFFT_F32 && c->precision != MIFFT_F64) return set_err(MIFFT_E_INVALID, "aux_copy: bad precision");
    for (int d = 0; d < c->ndim; ++d)
        if (c->dims[d] < 1) return set_err(MIFFT_E_INVALID, "aux_copy: dims[%d] = %lld", d, (long long)c->dims[d]);
    if (!src0 || !dst0 || (c->src_split && !src1) || (c->dst_split && !dst1)) return set_err(MIFFT_E_INVALID, "aux_copy: null buffer");
    const int rc = mifft_aux_copy_launch(c, src0, src1, dst0, dst1, (hipStream_t)stream);
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_aux_mul_rows(int32_t precision, void* a, const void* b, int64_t rows, int64_t n, mifft_stream_t stream) {
    if (!a || !b || rows < 0 || n < 1) return set_err(MIFFT_E_INVALID, "aux_mul_rows: bad arguments");
    const int rc = mifft_aux_mul_rows_launch(precision == MIFFT_F64, a, b, rows, n, (hipStream_t)stream);
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_aux_count_mismatch(const void* a, const void* b, size_t nbytes, uint64_t* count, mifft_stream_t stream) {
    if (!a || !b || !count) return set_err(MIFFT_E_INVALID, "null argument");
    if ((nbytes & 15) || (((uintptr_t)a | (uintptr_t)b) & 15) || ((uintptr_t)count & 7))
        return set_err(MIFFT_E_INVALID, "mifft_aux_count_mismatch: buffers and size in whole 16-byte words, an 8-byte aligned counter");
    const int rc = mifft_aux_mismatch_launch(a, b, (unsigned long long)(nbytes / 16), (unsigned long long*)count, (hipStream_t)stream);
    return rc ? hip_check((hipError_t)rc, "kernel launch") : 0;
}

int mifft_mixed_supported(int32_t precision, int32_t n) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    return mifft_mixed_supported_impl(precision == MIFFT_F64, n) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_launch_mixed_rows(int32_t precision, int32_t n, int64_t rows, int64_t stride_in, int64_t stride_out, const void* in, void* out,
                            const void* tw, int32_t inverse, double scale, mifft_stream_t stream) {
    if (mifft_mixed_supported(precision, n) != 0) return set_err(MIFFT_E_UNSUPPORTED, "mixed rows: no kernel for n = %d", n);
    if (!in || !out || !tw) return set_err(MIFFT_E_INVALID, "mixed rows: null buffer");
    if (rows < 0 || stride_in < n || stride_out < n) return set_err(MIFFT_E_INVALID, "mixed rows: bad row count / stride");
    if (const char* why = check_rows(precision, in, out, rows, n, stride_in, stride_out)) return set_err(MIFFT_E_INVALID, "mixed rows: %s", why);
    if (rows == 0) return 0;
    // dense rows: the single-buffer tile kernel of fft_mixed_nd.hip (one axis) where it is the faster one (development switch
    // MIFFT_DEBUG_ROWS_ND: 1 never, 2 wherever it fits)
    if (stride_in == n && stride_out == n && g_debug[MIFFT_DEBUG_ROWS_ND] != 1 && mifft_mixed_nd_rows_ok_impl(precision == MIFFT_F64, n) == 0 &&
        (g_debug[MIFFT_DEBUG_ROWS_ND] == 2 || mixed_rows_prefer_nd(precision == MIFFT_F64, n))) {
        const int rcn = mifft_mixed_nd_launch(precision == MIFFT_F64, n, 1, 1, rows, in, out, tw, nullptr, nullptr, inverse ? 3 : 0, scale, (hipStream_t)stream);
        if (rcn == -1) return set_err(MIFFT_E_INVALID, "grid too large");
        if (rcn != 0 && rcn != -2) return hip_check((hipError_t)rcn, "kernel launch");
        if (rcn == 0) return 0;
    }
    const int rc = mifft_mixed_launch(precision == MIFFT_F64, n, rows, stride_in, stride_out, 1, in, out, tw, inverse ? 3 : 0, scale, (hipStream_t)stream);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_launch_mixed_lines(int32_t precision, int32_t n, int64_t outer, int64_t inner, const void* in, void* out, const void* tw,
                             int32_t conj_in, int32_t conj_out, double scale, mifft_stream_t stream) {
    if (mifft_mixed_supported(precision, n) != 0) return set_err(MIFFT_E_UNSUPPORTED, "mixed lines: no kernel for n = %d", n);
    if (!in || !out || !tw) return set_err(MIFFT_E_INVALID, "mixed lines: null buffer");
    if (outer < 0 || inner < 1) return set_err(MIFFT_E_INVALID, "mixed lines: bad index space");
    if (mul3_checked(outer, inner, (long long)n * (precision == MIFFT_F64 ? 16 : 8)) < 0)
        return set_err(MIFFT_E_INVALID, "mixed lines: outer * inner * n overflows");
    if (const char* why = check_rows(precision, in, out, 0, n, n, n)) return set_err(MIFFT_E_INVALID, "mixed lines: %s", why);
    if (outer == 0) return 0;
    const int flags = (conj_in ? 1 : 0) | (conj_out ? 2 : 0);
    const int rc = mifft_mixed_launch(precision == MIFFT_F64, n, outer * inner, n, n, inner, in, out, tw, flags, scale, (hipStream_t)stream);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_mixed_long_split(int32_t precision, int64_t n, int32_t* n1, int32_t* n2) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    if (!n1 || !n2) return set_err(MIFFT_E_INVALID, "mixed long: null result pointer");
    int a = 0, b = 0;
    if (mifft_mixed_long_split_impl(precision == MIFFT_F64, n, &a, &b) != 0) return MIFFT_E_UNSUPPORTED;
    *n1 = a; *n2 = b;
    return 0;
}

int mifft_launch_mixed_long(int32_t precision, int32_t n1, int32_t n2, int64_t batch, const void* in, void* mid, void* out, const void* tw1,
                            const void* tw2, const void* tw_lo, const void* tw_hi, int32_t tw_shift, int32_t inverse, double scale,
                            mifft_stream_t stream) {
    if (mifft_mixed_supported(precision, n1) != 0 || mifft_mixed_supported(precision, n2) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "mixed long: no kernel for %d x %d", n1, n2);
    if (!in || !mid || !out || !tw1 || !tw2 || !tw_lo || !tw_hi) return set_err(MIFFT_E_INVALID, "mixed long: null buffer");
    if (mid == in) return set_err(MIFFT_E_INVALID, "mixed long: the first pass transposes, `mid` must not be the input");
    if (batch < 0 || tw_shift < 1 || tw_shift > 23) return set_err(MIFFT_E_INVALID, "mixed long: bad batch / table shift");
    if ((long long)n1 * n2 > (1ll << 24)) return set_err(MIFFT_E_INVALID, "mixed long: n1 * n2 = %lld exceeds 2^24", (long long)n1 * n2);
    if (mul3_checked(batch, (long long)n1 * n2, precision == MIFFT_F64 ? 16 : 8) < 0) return set_err(MIFFT_E_INVALID, "mixed long: batch * n overflows");
    {
        const uintptr_t mask = precision == MIFFT_F64 ? 15 : 7;
        if (((uintptr_t)in | (uintptr_t)mid | (uintptr_t)out) & mask) return set_err(MIFFT_E_INVALID, "mixed long: data buffers must be aligned to one complex number");
    }
    if (batch == 0) return 0;
    const int rc = mifft_mixed_long_launch(precision == MIFFT_F64, n1, n2, batch, in, mid, out, tw1, tw2, tw_lo, tw_hi, tw_shift,
                                           inverse ? 3 : 0, scale, (hipStream_t)stream);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_bluestein_padded(int32_t precision, int32_t n, int32_t* m) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    if (!m) return set_err(MIFFT_E_INVALID, "bluestein: null result pointer");
    const int v = mifft_bluestein_padded_impl(precision == MIFFT_F64, n);
    if (v <= 0) return MIFFT_E_UNSUPPORTED;
    *m = v;
    return 0;
}

int mifft_launch_bluestein_rows(int32_t precision, int32_t n, int32_t m, int64_t rows, int64_t stride_in, int64_t stride_out, const void* in,
                                void* out, const void* tw, const void* chirp, const void* bhat, int32_t inverse, double scale,
                                mifft_stream_t stream) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    if (n < 2 || m < 2 * (int64_t)n - 1 || mifft_bluestein_len_supported_impl(precision == MIFFT_F64, m) != 0)
        return set_err(MIFFT_E_UNSUPPORTED, "bluestein rows: no kernel for n = %d padded to %d", n, m);
    if (!in || !out || !tw || !chirp || !bhat) return set_err(MIFFT_E_INVALID, "bluestein rows: null buffer");
    if (rows < 0 || stride_in < n || stride_out < n) return set_err(MIFFT_E_INVALID, "bluestein rows: bad row count / stride");
    if (const char* why = check_rows(precision, in, out, rows, n, stride_in, stride_out)) return set_err(MIFFT_E_INVALID, "bluestein rows: %s", why);
    if (rows == 0) return 0;
    const int rc = mifft_bluestein_launch(precision == MIFFT_F64, n, m, rows, stride_in, stride_out, in, out, tw, chirp, bhat,
                                          inverse ? 3 : 0, scale, (hipStream_t)stream);
    if (rc == -2) return set_err(MIFFT_E_UNSUPPORTED, "bluestein rows: no kernel for n = %d padded to %d", n, m);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_mixed_nd_supported(int32_t precision, int32_t x, int32_t y, int32_t z) {
    if (precision != MIFFT_F32 && precision != MIFFT_F64) return MIFFT_E_UNSUPPORTED;
    return mifft_mixed_nd_supported_impl(precision == MIFFT_F64, x, y, z) == 0 ? 0 : MIFFT_E_UNSUPPORTED;
}

int mifft_launch_mixed_nd(int32_t precision, int32_t x, int32_t y, int32_t z, int64_t transforms, const void* in, void* out, const void* tw_x,
                          const void* tw_y, const void* tw_z, int32_t inverse, double scale, mifft_stream_t stream) {
    if (mifft_mixed_nd_supported(precision, x, y, z) != 0) return set_err(MIFFT_E_UNSUPPORTED, "mixed nd: no kernel for %d x %d x %d", z, y, x);
    if (!in || !out || (x > 1 && !tw_x) || (y > 1 && !tw_y) || (z > 1 && !tw_z)) return set_err(MIFFT_E_INVALID, "mixed nd: null buffer");
    if (const char* why = check_rows(precision, in, out, 0, 0, 0, 0)) return set_err(MIFFT_E_INVALID, "mixed nd: %s", why);
    if (transforms < 0 || mul3_checked(transforms, (long long)x * y * z, precision == MIFFT_F64 ? 16 : 8) < 0)
        return set_err(MIFFT_E_INVALID, "mixed nd: bad transform count");
    if (transforms == 0) return 0;
    // inverse: 0 forward, 1 inverse = conjugate on load and on store; 2 / 4: on load / on store only (one end of a composition)
    const int flags = inverse == 1 ? 3 : inverse == 2 ? 1 : inverse == 4 ? 2 : 0;
    if (inverse != 0 && inverse != 1 && inverse != 2 && inverse != 4) return set_err(MIFFT_E_INVALID, "mixed nd: inverse must be 0, 1, 2 or 4");
    const int rc = mifft_mixed_nd_launch(precision == MIFFT_F64, x, y, z, transforms, in, out, tw_x, tw_y, tw_z, flags, scale, (hipStream_t)stream);
    if (rc == -2) return set_err(MIFFT_E_UNSUPPORTED, "mixed nd: no kernel for %d x %d x %d", z, y, x);
    if (rc == -1) return set_err(MIFFT_E_INVALID, "grid too large");
    if (rc != 0) return hip_check((hipError_t)rc, "kernel launch");
    return 0;
}

int mifft_time_chain(const mifft_pass* passes, int32_t npasses, void* const bufs0[3], void* const bufs1[3], mifft_stream_t stream,
                     int32_t repeats, float* ms_total) {
    if (!ms_total || repeats < 1) return set_err(MIFFT_E_INVALID, "bad timing arguments");
    hipEvent_t e0, e1;
    int rc = hip_check(hipEventCreate(&e0), "hipEventCreate");
    if (rc) return rc;
    rc = hip_check(hipEventCreate(&e1), "hipEventCreate");
    if (rc) {
        (void)hipEventDestroy(e0);
        return rc;
    }
    rc = hip_check(hipEventRecord(e0, (hipStream_t)stream), "hipEventRecord");
    for (int r = 0; r < repeats && !rc; ++r) rc = mifft_launch_chain(passes, npasses, bufs0, bufs1, stream);
    if (!rc) rc = hip_check(hipEventRecord(e1, (hipStream_t)stream), "hipEventRecord");
    if (!rc) rc = hip_check(hipEventSynchronize(e1), "hipEventSynchronize");
    if (!rc) rc = hip_check(hipEventElapsedTime(ms_total, e0, e1), "hipEventElapsedTime");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
