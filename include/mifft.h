/*
 * mifft.h -- C ABI of libmifft.so, the MI355X (gfx950) batched power-of-two c2c FFT engine
 * behind pyfft's Plan()/execute() API.
 *
 * This is the drop-in boundary.  In the reference (fjarri-attic/pyfft 0.3.9) the host side
 * (pyfft/plan.py, pyfft/kernel.py) reaches the device through a duck-typed
 * Context / Module / Function protocol implemented on PyCUDA (pyfft/cuda.py:19-113):
 *
 *   context.allocate(nbytes)                          cuda.py:85-89     -> mifft_malloc / mifft_free
 *   context.compile(src, fast_math) -> Module         cuda.py:52-61,91  -> (none: kernels are AOT-compiled
 *   module.getFunction(name, split, block) -> Function cuda.py:19-30        into this library; a pass is
 *   func.isExecutable()                               cuda.py:48-49         selected by mifft_pass_supported)
 *   func.prepare(grid, S); func(stream, *ptrs)        cuda.py:32-46     -> mifft_launch_pass / mifft_launch_chain
 *   context.createQueue()/wait()/flush()/getQueue()   cuda.py:94-107    -> mifft_stream_create / _sync / _destroy
 *   device limits read in Context.__init__            cuda.py:72-83     -> mifft_device_props
 *
 * and the generated kernels have the signature fft{Fwd,Inv}(in, out, int S) or
 * (in_re, in_im, out_re, out_im, int S) (pyfft/kernel.mako:690-697).  A `mifft_pass` describes one
 * such launch: one Stockham pass of pyfft/kernel.mako:805-1047 (globalKernel) or a whole LDS-resident
 * transform along the contiguous axis (kernel.mako:725-803, localKernel).
 *
 * Conventions
 *   - plain C, no C++ types or exceptions cross the boundary;
 *   - every function returns 0 on success, a positive hipError_t, or a negative MIFFT_E_* code, and
 *     records a thread-local message readable with mifft_last_error();
 *   - all device buffers (user data, temp, twiddle tables) are owned by the caller and only borrowed for
 *     the duration of the enqueued work; the library keeps no per-plan or per-call state (process-wide are only the
 *     DEFAULTS of the development switches, mifft_debug_set_default; mifft_debug_set itself is per thread);
 *   - launches are asynchronous on the caller's stream; nothing here synchronises except the *_sync calls.
 */
#ifndef MIFFT_H
#define MIFFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: mifft_launch_fused2's counter buffer is MIFFT_FUSED2_COUNTER_BYTES (one 256-byte line per counter) and is zeroed by the
 * call; streams are blocking streams; mifft_stream_wait_event
 * 3: the persistent launches take a mifft_fused_sync: two alternating counter sets, error word anywhere the device can write;
 * mifft_device_props carries the last-level cache size and the XCD count; mifft_launch_fused_pair */
#define MIFFT_ABI_VERSION 5

/* negative library error codes (positive values are hipError_t) */
#define MIFFT_E_INVALID      (-1)  /* malformed descriptor / argument              */
#define MIFFT_E_UNSUPPORTED  (-2)  /* no compiled kernel for this (precision, L, mode) */
#define MIFFT_E_NODEVICE     (-3)  /* no HIP device visible                         */

/* precision of the arithmetic and of the buffers */
#define MIFFT_F32 0
#define MIFFT_F64 1

/* buffer layout: interleaved (re,im) pairs, or two scalar planes
 * (pyfft/plan.py:26-35: complex dtypes -> interleaved, float dtypes -> split) */
#define MIFFT_INTERLEAVED 0
#define MIFFT_SPLIT       1

/* mifft_pass.flags (only meaningful with layout == MIFFT_SPLIT): the buffer read / written by this pass is
 * interleaved although the plan's layout is split.  Used for the plan-owned temp buffer, which is always
 * interleaved: 16 columns of one fp32 plane are only 64 bytes per row, so every side kept interleaved streams
 * at the full rate (src/dst plane pointer of that side is ignored and may be NULL). */
#define MIFFT_FLAG_SRC_INTERLEAVED 1
#define MIFFT_FLAG_DST_INTERLEAVED 2
/* hints for multi-pass plans: the source of this pass is read once and not needed again (the plan's first pass) /
   the destination is not re-read by the plan (its last pass): kernels that honour them use non-temporal accesses,
   which leaves the Infinity Cache to the intermediate of the pipelined strategy.  Results are unaffected. */
#define MIFFT_FLAG_STREAM_SRC 4
#define MIFFT_FLAG_STREAM_DST 8
/* this pass and the NEXT one of the chain are run by one launch (a "pass pair", csrc/fft_pair.hpp): the launch reads the
   buffer `src` of this pass and writes the buffer `dst` of the next one; the chain launchers skip the second descriptor.
   Only set where mifft_pass_pair_supported() says so. */
#define MIFFT_FLAG_PAIR_WITH_NEXT 16
/* small launches: store the destination WRITE-THROUGH.  A kernel's plain stores stay dirty in the eight L2s (32 MiB) until the
   end-of-kernel write-back, which runs alone: ~4 us behind a launch whose whole output fits the L2s -- a third of a 32 MiB
   launch (profiles/r03_l_small_launch_write_through.log).  Set by the planner for launches of <= 128 MiB; kernels without a
   write-through form ignore it.  Results are unaffected. */
#define MIFFT_FLAG_WRITE_THROUGH 32

/* pass kinds */
#define MIFFT_PASS_COL 0  /* strided pass: [outer][L][M*S] -> [outer][M][L][S]  (kernel.mako:805-1047) */
#define MIFFT_PASS_ROW 1  /* contiguous pass: `outer` rows of L points, in place  (kernel.mako:725-803)  */
#define MIFFT_PASS_ND  2  /* whole small 2-D/3-D transform in LDS: L = x, M = y, S = z (x contiguous), `outer`
                            transforms back to back; tw_L / tw_lo / tw_hi = w(x)^k / w(y)^k / w(z)^k tables (NULL for an
                            axis of length 1); shape accepted by mifft_nd_shape_supported().  In place capable.
                            Replaces the reference's local kernel + one global chain per further axis (plan.py:111-123) */

typedef void *mifft_stream_t; /* hipStream_t; NULL = the default stream */
typedef void *mifft_event_t;  /* hipEvent_t */

/*
 * One launch.  With w(m) = exp(-2*pi*i/m) (forward; the inverse conjugates input and output):
 *
 *   MIFFT_PASS_COL  in  viewed as [outer][L][M][S]
 *                   out viewed as [outer][M][L][S]
 *       out[o][l][q][j] = scale * w(L*M)^(l*q) * sum_r in[o][r][l][j] * w(L)^(r*q)
 *     M == 1  -> plain strided transform (last pass of an axis), in-place capable;
 *     S == 1  -> first pass of a long contiguous axis: the write is a transposition.
 *
 *   MIFFT_PASS_ROW  in/out viewed as [outer][L]:  out[o][q] = scale * sum_r in[o][r] * w(L)^(r*q)
 *
 *   MIFFT_PASS_ND   in/out viewed as [outer][S][M][L] (z, y, x):  out[o] = scale * DFT3(in[o])
 *
 * Matrix `o` starts at element o*outer_stride_{in,out} (elements of the complex type; for split layout
 * the same offset applies to both planes).  Sizes are powers of two except `outer`.
 */
typedef struct mifft_pass {
    int32_t kind;        /* MIFFT_PASS_COL | MIFFT_PASS_ROW | MIFFT_PASS_ND */
    int32_t precision;   /* MIFFT_F32 | MIFFT_F64 */
    int32_t layout;      /* MIFFT_INTERLEAVED | MIFFT_SPLIT (same for input and output) */
    int32_t inverse;     /* 0 forward (numpy.fft.fft sign), 1 inverse (unnormalised unless `scale`) */
    int32_t L;           /* transform length of this launch (the radix of the pass) */
    int32_t variant;     /* kernel variant selector: 0 = library default; 1 = the generic kernel of the kind (COL / ROW: the LDS tile kernel; ND: the
                          * run-time-shaped kernel instead of the shape's fixed instance -- the plan's choice for the shapes listed in the tuning table) */
    int64_t M;           /* not-yet-transformed extent of the axis after this pass (COL), 1 for ROW */
    int64_t S;           /* extent of everything faster than the digit being transformed (COL), 1 for ROW */
    int64_t outer;       /* number of independent matrices / rows */
    int64_t outer_stride_in;   /* elements between consecutive matrices in the input  */
    int64_t outer_stride_out;  /* elements between consecutive matrices in the output */
    double  scale;       /* every output is multiplied by this (1.0 except in a plan's last pass,
                            pyfft/kernel.py:23-37, pyfft/plan.py:125-128) */
    const void *tw_L;    /* device: L entries w(L)^k, complex of `precision` */
    const void *tw_lo;   /* device: 2^tw_shift entries w(L*M)^k            (NULL when M == 1) */
    const void *tw_hi;   /* device: (L*M)>>tw_shift entries w(L*M)^(k<<tw_shift) (NULL when M == 1) */
    int32_t tw_shift;
    int32_t src;         /* for mifft_launch_chain: index of the buffer read  (0 in, 1 out, 2 temp) */
    int32_t dst;         /* for mifft_launch_chain: index of the buffer written */
    int32_t flags;       /* MIFFT_FLAG_*: per-side layout overrides for a plan's internal temp buffer */
} mifft_pass;

typedef struct mifft_device_props {
    char    name[256];
    char    gcn_arch[64];
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t max_threads_per_block;
    int32_t max_grid_x;
    int64_t lds_bytes_per_block;
    int64_t total_mem_bytes;
    int32_t clock_khz;
    int32_t l2_bytes;        /* L2 of ONE XCD (hipDeviceProp_t::l2CacheSize) */
    int64_t llc_bytes;       /* last-level cache in front of HBM (the Infinity Cache / MALL: 256 MiB on MI355X) from the HSA agent
                                (HSA_AGENT_INFO_CACHE_SIZE[2]); 0 = none or unknown.  The planner sizes the rings of the persistent
                                launches, the pipelined chunks and the write-through rule from it instead of from literals */
    int32_t num_xcc;         /* XCDs (chiplets with their own L2) the device spans (HSA_AMD_AGENT_INFO_NUM_XCC); 1 if unknown */
    int32_t reserved0;
} mifft_device_props;

/* ---- library ------------------------------------------------------------------------------------- */
int         mifft_abi_version(void);
const char *mifft_last_error(void);
/* development switches (all 0 in production; pyfft_amd/_debug.py maps environment variables onto them) */
#define MIFFT_DEBUG_NO_ND2 0       /* run-time-shaped N-D kernel only */
#define MIFFT_DEBUG_FUSED_NO_NT 1  /* fused two-pass kernel without non-temporal hints */
#define MIFFT_DEBUG_NO_WAVE 2      /* no wave-autonomous small-transform kernels */
#define MIFFT_DEBUG_FORCE_WAVE 3   /* wave-autonomous kernels wherever one exists, whatever the buffer size */
#define MIFFT_DEBUG_PERSIST 4      /* persistent (prefetching) form of the long fp32 rows (measured: no gain) */
#define MIFFT_DEBUG_ALT_ROWS 5     /* alternative stage lists of the longest fp32 rows (A/B measurements) */
#define MIFFT_DEBUG_PAIR 6         /* pass pairs: 0 = default split, 1 = off, 2 = the alternative y split, 3 = the first tile forms of the persistent two-pair kernel (A/B measurements) */
#define MIFFT_DEBUG_STORE 7        /* streamed output stores (A/B): 0 = default, 1 = non-temporal, 2 = write-through (sc1), 3 = plain */
#define MIFFT_DEBUG_ROWS_ND 8      /* dense smooth rows: 0 = default, 1 = two-buffer row kernel only, 2 = single-buffer tile kernel wherever it fits (A/B) */
#define MIFFT_DEBUG_NARROW_TILES 9 /* A/B of the round-4 tile forms: 1 = the rounds 1-3 forms -- fp32 L = 256 / 512 on 16-column tiles also in the persistent
                                    * kernel, and for split-complex planes no lane-interleaved double tiles, no register-edged rows, no fixed-shape N-D
                                    * route, no row-first 2-D kernel, no write-through in the run-time-shaped N-D kernel; 2 = 32-column tiles (and the
                                    * double tile of a plane-writing L = 1024 pass) also in plain launches; 3 = 16-column tiles also for strided passes
                                    * whose rows lie >= 2^16 points apart (round 6: those run on 32-column tiles by default) */
#define MIFFT_DEBUG_NO_ROWFIRST 10 /* split-complex fp32 2-D persistent launches: 1 = two transposing passes on sibling tiles instead of the row-first kernel (A/B) */
#define MIFFT_DEBUG_PREFETCH 11 /* persistent kernels on 512-thread tiles, `make DEV=1` builds: 1 = the work list that issues a tile's loads before the publish of
                                  * the previous one (round 6; measured equal to the round-2 list: 0.392 / 0.393 on configuration 5) */
#define MIFFT_DEBUG_KEYS 12
/* mifft_debug_set changes a switch for the CALLING THREAD only (a thread that never set a key sees the process default), so a
 * measurement that flips a switch in one thread cannot change the kernels another thread's plan gets; mifft_debug_set_default sets
 * the process default (what the environment variables of pyfft_amd/_debug.py do once, when the library is loaded).  A switch must
 * not change between the support queries a plan is built from and that plan's launches: the launch then fails with
 * MIFFT_E_UNSUPPORTED instead of running another kernel. */
int mifft_debug_set(int32_t key, int32_t value);
int mifft_debug_set_default(int32_t key, int32_t value);
int mifft_debug_get(int32_t key);
/* Optional parts of the library.  The default build leaves out the measured-and-not-adopted strategies (`make DEV=1` builds them):
 * their launchers then return MIFFT_E_UNSUPPORTED and mifft_has_feature says so beforehand. */
#define MIFFT_FEATURE_XCD2 0            /* mifft_launch_xcd2: XCD-resident single-crossing form of 1024 x 1024 (0.32-0.34 against 0.44) */
#define MIFFT_FEATURE_FUSED2X 1         /* mifft_launch_fused2x: one work list per XCD */
#define MIFFT_FEATURE_SEQUENTIAL_LIST 2 /* lag == 0 in the persistent launchers: both passes of a tiny batch in one launch */
#define MIFFT_FEATURE_AB_FORMS 3        /* the kernel forms only the development switches select (A/B measurements): 16-column tiles for
                                         * every length and unpaired plane tiles in the persistent kernels (MIFFT_DEBUG_NARROW_TILES = 1),
                                         * plain / write-through streams (MIFFT_DEBUG_FUSED_NO_NT, MIFFT_DEBUG_STORE), persistent and
                                         * alternative long rows (MIFFT_DEBUG_PERSIST, MIFFT_DEBUG_ALT_ROWS), the two-transposing-pass form
                                         * of split-complex 2-D squares; without it such a request fails with MIFFT_E_UNSUPPORTED */
int mifft_has_feature(int32_t feature); /* 1 = built in, 0 = not */

/* ---- runtime shim (replaces cuda.py Context: allocate / stream lifecycle / device limits) ---------- */
int mifft_device_count(int *count);
int mifft_set_device(int device);
int mifft_get_device(int *device);
int mifft_device_props_get(int device, mifft_device_props *props);
int mifft_malloc(void **ptr, size_t nbytes);
int mifft_free(void *ptr);
int mifft_memset(void *ptr, int value, size_t nbytes, mifft_stream_t stream);
int mifft_memcpy_h2d(void *dst, const void *src, size_t nbytes, mifft_stream_t stream);
int mifft_memcpy_d2h(void *dst, const void *src, size_t nbytes, mifft_stream_t stream);
int mifft_memcpy_d2d(void *dst, const void *src, size_t nbytes, mifft_stream_t stream);
/* pinned host memory + a copy that does NOT synchronise: how a plan reads a persistent kernel's error word without
 * stalling the stream (the copy is enqueued behind the launch, the host looks at it once mifft_event_query says so) */
int mifft_host_alloc(void **ptr, size_t nbytes);
int mifft_host_free(void *ptr);
int mifft_memcpy_d2h_async(void *dst, const void *src, size_t nbytes, mifft_stream_t stream);
/* streams are BLOCKING streams (they order against the legacy null stream, like PyCUDA's: cuda.py:94-96) */
int mifft_stream_create(mifft_stream_t *stream);
int mifft_stream_destroy(mifft_stream_t stream);
int mifft_stream_sync(mifft_stream_t stream);
int mifft_device_sync(void);
/*
 * Stream capture / hipGraph replay of the calls a plan enqueues (HIP graphs are how a launch-bound inner loop is replayed on this
 * part; the reference's execute is an asynchronous enqueue on the caller's stream, pyfft/plan.py:250-259, and a torch-ROCm caller
 * may capture that stream).  Every launcher of this library may be called on a capturing stream; the persistent launches then
 * REQUIRE the single-set form of their mifft_fused_sync argument -- counters_next == NULL --, see there.
 *   mifft_stream_is_capturing   *capturing = 1 while `stream` records into a graph
 *   mifft_stream_begin_capture  start recording the work enqueued on `stream` (relaxed mode; not the default stream)
 *   mifft_stream_end_capture    stop recording and instantiate: *graph is an executable graph (hipGraphExec_t)
 *   mifft_graph_launch          replay it on `stream`; mifft_graph_destroy frees it
 */
typedef void *mifft_graph_t; /* hipGraphExec_t */
int mifft_stream_is_capturing(mifft_stream_t stream, int32_t *capturing);
int mifft_stream_begin_capture(mifft_stream_t stream);
int mifft_stream_end_capture(mifft_stream_t stream, mifft_graph_t *graph);
int mifft_graph_launch(mifft_graph_t graph, mifft_stream_t stream);
int mifft_graph_destroy(mifft_graph_t graph);
int mifft_event_create(mifft_event_t *event);
int mifft_event_destroy(mifft_event_t event);
int mifft_event_record(mifft_event_t event, mifft_stream_t stream);
int mifft_event_sync(mifft_event_t event);
int mifft_event_query(mifft_event_t event); /* 0 = completed, 1 = not yet, else an error code */
/* work enqueued on `stream` after this call starts only when `event` has completed (hipStreamWaitEvent): how a plan whose
 * caller switches streams between asynchronous executes orders the second stream behind the first on its scratch */
int mifft_stream_wait_event(mifft_stream_t stream, mifft_event_t event);
int mifft_event_elapsed_ms(float *ms, mifft_event_t start, mifft_event_t stop);

/* ---- pass launchers (replace cuda.py Function.__call__, cuda.py:35-46) ----------------------------- */

/* largest x*y*z a MIFFT_PASS_ND launch accepts for ANY power-of-two shape and either layout of the precision */
int mifft_nd_max_points_for(int32_t precision);

/* 0 if one MIFFT_PASS_ND launch can transform the (z, y, x) shape, else MIFFT_E_UNSUPPORTED: every shape up to
 * mifft_nd_max_points_for(); with variant MIFFT_VARIANT_INTERLEAVED_ONLY also the larger fixed shapes that exist for
 * interleaved data on both sides only; with MIFFT_VARIANT_SPLIT_ONLY those that exist for split-complex planes on BOTH
 * sides (a single-pass plan of a split-complex layout); with MIFFT_VARIANT_OUT_OF_PLACE_ONLY (round 5) those whose kernel runs several
 * work-groups per transform (csrc/fft_nd2z.hpp): interleaved data on both sides AND input != output -- a caller keeps another chain for
 * its in-place executes.  (No counterpart in the reference, whose kernels are generated per plan.) */
#define MIFFT_VARIANT_SPLIT_ONLY 3
#define MIFFT_VARIANT_OUT_OF_PLACE_ONLY 4
#define MIFFT_VARIANT_OUT_OF_PLACE_ANY_SIZE 5 /* ... and that kernel is the better choice at EVERY buffer size (else: beyond half the last-level cache per side) */
#define MIFFT_VARIANT_SPLIT_OUT_OF_PLACE 6 /* round 6: the same for split-complex planes on both sides (csrc/fft_nd2zp.hpp) ... */
#define MIFFT_VARIANT_SPLIT_OUT_OF_PLACE_ANY_SIZE 7 /* ... and at every buffer size */
int mifft_nd_shape_supported(int32_t precision, int32_t x, int32_t y, int32_t z, int32_t variant);

/* 0 if a compiled kernel exists for (kind, precision, L, variant), else MIFFT_E_UNSUPPORTED.
 * Counterpart of Function.isExecutable (cuda.py:48-49) for AOT kernels.  variant 0: a kernel that takes either layout;
 * MIFFT_VARIANT_INTERLEAVED_ONLY (ROW passes): also count kernels that need interleaved data on both sides.
 * The query knows no layout: the longest rows (32768 points fp32, 16384 fp64) exist for interleaved -> interleaved, planes -> planes and
 * planes -> interleaved; an interleaved -> planes ROW pass of that length (MIFFT_FLAG_SRC_INTERLEAVED without _DST_INTERLEAVED on a
 * MIFFT_SPLIT pass) is refused by mifft_launch_pass with MIFFT_E_UNSUPPORTED although the query says 0 -- no plan builds one (a ROW pass
 * is a plan's first pass: its input is the user's layout). */
#define MIFFT_VARIANT_INTERLEAVED_ONLY 2
int mifft_pass_supported(int32_t kind, int32_t precision, int32_t L, int32_t variant);

/* Enqueue one pass.  in1/out1 are the imaginary planes for MIFFT_SPLIT and must be NULL for
 * MIFFT_INTERLEAVED.  In-place (out == in) is allowed for MIFFT_PASS_ROW, MIFFT_PASS_ND and for
 * MIFFT_PASS_COL with M == 1 (pyfft/kernel.py:144,238-241). */
int mifft_launch_pass(const mifft_pass *pass, const void *in0, const void *in1, void *out0, void *out1,
                      mifft_stream_t stream);

/*
 * Pass pairs: two CONSECUTIVE passes of a chain run by one launch on tiles that hold the points of both, so that a 3-D
 * transform whose (y, x) plane fits no work-group still crosses HBM twice instead of three times.  With the y axis
 * factored R0 * R1 as the chain factors a long axis (pyfft/kernel.py:259-283), the four passes
 *     ROW x | COL y (L = R0, M = R1, S = nx) | COL y (L = R1, M = 1, S = nx * R0) | COL z (L = nz, M = 1, S = nx * ny)
 * are the pairs (ROW x, COL y R0) -- out of place only -- and (COL y R1, COL z) -- in place capable.  Dense batches (outer
 * stride = the transform size) only.  The buffer BETWEEN the two launches is interleaved; with layout MIFFT_SPLIT the first
 * launch reads and the second writes two scalar planes (flags MIFFT_FLAG_DST_INTERLEAVED / _SRC_INTERLEAVED on the inner side).
 *   mifft_pair_split          R0 (> 0) if the library has both pair kernels for a (z, y, x) transform of that layout, else 0
 *   mifft_pass_pair_supported 0 if (p0, p1) is such a pair with a compiled kernel, else MIFFT_E_UNSUPPORTED
 *   mifft_launch_pass_pair    enqueue it: reads in0 / in1 (p0's input side), writes out0 / out1 (p1's output side); in1 / out1
 *                             are the imaginary planes of a split side, else ignored; scale = p0->scale * p1->scale
 */
int mifft_pair_split(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z);
int mifft_pass_pair_supported(const mifft_pass *p0, const mifft_pass *p1);
/* Round 5: 0 if the library has the pass-pair kernel itself -- kind 0 = (ROW x, COL y R0) keyed by (nx, R0, R1 = ny / R0), kind 1 =
 * (COL y R1, COL z) keyed by (S0 = what is faster than the y digit, R1, nz) -- for interleaved data (layout MIFFT_SPLIT: planes on the
 * user side of the pair).  What a planner asks before it builds a chain around ONE pair: a 2-D shape with a y axis too long for one
 * strided pass (pair XY + one plain pass instead of three launches), a 3-D shape with short y and z behind a long x (ROW x + pair YZ). */
int mifft_pair_kernel_supported(int32_t precision, int32_t layout, int32_t kind, int32_t k0, int32_t k1, int32_t k2);
int mifft_launch_pass_pair(const mifft_pass *p0, const mifft_pass *p1, const void *in0, const void *in1, void *out0, void *out1,
                           mifft_stream_t stream);

/* Enqueue a whole plan: passes[i] reads bufs0/bufs1[passes[i].src] and writes [passes[i].dst]
 * (0 = data_in, 1 = data_out, 2 = temp -- the ping-pong loop of pyfft/plan.py:217-248 with the
 * schedule already decided by the Python plan).  bufs1 may be NULL for interleaved layout. */
int mifft_launch_chain(const mifft_pass *passes, int32_t npasses, void *const bufs0[3], void *const bufs1[3],
                       mifft_stream_t stream);

/*
 * Pipelined form of mifft_launch_chain for any multi-pass plan (counterpart of the batch loop the
 * reference runs inside each kernel grid, pyfft/kernel.py:99-121, re-cut for the MI355X memory system):
 * the batch is processed in chunks of `chunk` items; chunk i runs the whole pass chain on side stream
 * i % nside with temp slot i % nside, so the inter-pass intermediate of a chunk is consumed while it is still
 * in the 256 MiB Infinity Cache and kernels of different chunks overlap each other's launch tails.
 *   - `passes` describe the FULL batch (outer = outer_per_item * batch); `item_elems` = elements per batch item
 *     (per plane for split layout);  bufs*[2] (temp), when the schedule uses it, must hold nside * chunk items.
 *   - ordering: side streams wait for everything enqueued on `stream` so far; `stream` waits for all side
 *     streams before the call's work counts as done.  events[0..nside] are caller-owned scratch events.
 *   - nside == 1 with side[0] == stream: the chunks run in order on `stream` itself, no fork, no join, no event is touched -- what to
 *     pass while `stream` is being captured (a linear graph; forked captures of this launch crashed hipGraphLaunch now and then).
 */
int mifft_launch_chain_pipelined(const mifft_pass *passes, int32_t npasses, void *const bufs0[3], void *const bufs1[3],
                                 int64_t batch, int64_t chunk, int64_t item_elems, mifft_stream_t stream,
                                 const mifft_stream_t *side, int32_t nside, const mifft_event_t *events);

/*
 * Fused form of a two-pass long contiguous axis N = p0->L * p1->L: both Stockham passes of all `p0->outer` transforms in
 * ONE persistent launch, pass 1 of transform t trailing pass 0 by `lag` transforms, with the intermediate in a scratch
 * ring of `ring_slots` transforms (ring_slots > lag) that stays in the Infinity Cache.
 *   shapes    fp32: p0->L, p1->L in {256, 512, 1024}, or 2048 x 2048 / 2048 x 1024; fp64: 1024 x 1024.
 *             1-D form: p0 = the transposing first pass (COL, S == 1, M == p1->L), p1 = the plain strided last pass
 *             (COL, M == 1, S == p0->L).  2-D form (the squares 512 / 1024 / 2048 in fp32, 1024 in fp64): p0 = the ROW
 *             pass, p1 = the strided COL pass of the plan.  Anything else: MIFFT_E_UNSUPPORTED.
 *   sync      counters + error word, see mifft_fused_sync below (one 256-byte line per counter, csrc/fft_fused2.hpp).  After
 *             completion a non-zero error word reports a dependency time-out (results invalid).
 *   ring      always interleaved (ring0; ring1 is ignored), also for split-plane in/out buffers.
 */
#define MIFFT_FUSED2_COUNTER_STRIDE 64u /* uint32 words between two counters */
#define MIFFT_FUSED2_COUNTER_BYTES(outer) ((size_t)MIFFT_FUSED2_COUNTER_STRIDE * 4u * (9u + 2u * (size_t)(outer)))
/*
 * Synchronisation state of one persistent launch (all caller-owned device-accessible memory):
 *   counters       MIFFT_FUSED2_COUNTER_BYTES(outer) bytes.  With counters_next == NULL the call zeroes them on `stream` in front
 *                  of the launch (hipMemsetAsync: ~5 us, what a 32 MiB execute cannot afford; on a CAPTURING stream a small kernel
 *                  of the library's own, i.e. a kernel node -- a memset node in front of the launch stopped zeroing once the process
 *                  built another plan under the HIP runtime PyTorch bundles, round 6).  The ONLY form allowed on a
 *                  capturing stream (a replayed graph runs on the same set every time): MIFFT_E_INVALID otherwise.
 *   counters_next  a second buffer of the same size: the caller guarantees that `counters` is ALL ZERO when the launch starts, and
 *                  THIS launch zeroes `counters_next` -- a plan that alternates between two sets (set A zeroes B, B zeroes A) after
 *                  one initial mifft_memset of both never pays the memset again.
 *   error_word     a uint32 the kernel sets (system-scope store) when a bounded dependency wait times out = results INVALID; any
 *                  address the device can write: pinned host memory from mifft_host_alloc (device-accessible under the same
 *                  pointer; the host then reads it without a copy), or NULL = word [1] of `counters` -- in the single-set form only:
 *                  with counters_next the next launch zeroes that word, so the two-set form REQUIRES an error word of its own
 *                  (MIFFT_E_INVALID otherwise).  A word of the caller's is never cleared by the library.
 */
typedef struct mifft_fused_sync {
    void *counters;
    void *counters_next;
    void *error_word;
} mifft_fused_sync;
/* lag == 0 selects the SEQUENTIAL work list for tiny batches (ring_slots == outer): every first-pass tile of every transform,
 * then every second-pass tile -- two dependent launches folded into one, without the launch gap and the end-of-kernel
 * write-back between them (the reference's own 32 MiB benchmark protocol, test/test_performance.py:11,22-30).  Development form
 * (MIFFT_FEATURE_SEQUENTIAL_LIST; measured slower than two launches): the list is dealt out statically, which is deadlock-free
 * only while EVERY work-group of the launch is resident -- the launcher caps the grid by the kernel's occupancy on this device,
 * but it cannot see other streams, processes or CU masks, so the device must be the caller's alone; otherwise the bounded waits
 * time out (~4 s per item), the error word is set and the output is invalid. */
int mifft_launch_fused2(const mifft_pass *p0, const mifft_pass *p1, const void *in0, const void *in1, void *out0,
                        void *out1, void *ring0, void *ring1, int32_t ring_slots, int32_t lag, const mifft_fused_sync *sync,
                        int32_t grid, mifft_stream_t stream);

/*
 * The same launch with one work list PER XCD (strategy `fusedx`, csrc/fft_fused2.hpp): XCD x owns the transforms x, x + 8, ... and
 * the ring slots [x * ring_slots, (x + 1) * ring_slots), so the ring holds 8 * ring_slots transforms.  A work-group whose own
 * list is exhausted drains the other XCDs' lists, so the result does not depend on where the work-groups land (the
 * intermediate is written write-through, like the global form's).  fp32 1-D pairs with p0->L >= p1->L in {256, 512, 1024};
 * in1 / out1 = the imaginary planes of split-complex user buffers (NULL for interleaved data; the ring is always interleaved).
 */
int mifft_launch_fused2x(const mifft_pass *p0, const mifft_pass *p1, const void *in0, const void *in1, void *out0, void *out1,
                         void *ring0, int32_t ring_slots, int32_t lag, const mifft_fused_sync *sync, int32_t grid,
                         mifft_stream_t stream);

/*
 * Persistent form of a 3-D plan made of two PASS PAIRS (mifft_pair_split > 0): passes[0..3] = ROW x | COL y (R0) | COL y (R1) |
 * COL z as the chain holds them.  Work list as in mifft_launch_fused2 with the (ROW x, COL y R0) tiles as first-pass items and the
 * (COL y R1, COL z) tiles as second-pass items; the buffer between the two pairs is a ring of `ring_slots` whole transforms
 * (interleaved) that stays in the last-level cache.  Exists for the cubes whose transform is a fraction of that cache:
 *   mifft_fused_pair_supported   0 if (precision, layout, x, y, z) has such a kernel, else MIFFT_E_UNSUPPORTED.  Split-complex user
 *                                buffers (in1 / out1 = the imaginary planes, layout MIFFT_SPLIT in the descriptors, the ring
 *                                interleaved: MIFFT_FLAG_DST_INTERLEAVED on passes[1], MIFFT_FLAG_SRC_INTERLEAVED on passes[2]):
 *                                the same shapes
 *   mifft_fused_pair_split       the factor R0 of the y axis that kernel is built for (y = R0 * R1), 0 if there is none.  For the
 *                                128^3 cubes it equals mifft_pair_split; the other shapes (64- and 128-point axes) have NO plain
 *                                pair launches -- their chain is a plane pass + a strided z pass -- and the caller builds the
 *                                four-pass list for this launch alone
 * Reference shape of the work: pyfft/plan.py:160-167 (one chain per axis), published row doc/source/index.rst:373 (128^3).
 */
int mifft_fused_pair_supported(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z);
int mifft_fused_pair_split(int32_t precision, int32_t layout, int32_t x, int32_t y, int32_t z);
int mifft_launch_fused_pair(const mifft_pass *passes, const void *in0, const void *in1, void *out0, void *out1, void *ring0,
                            int32_t ring_slots, int32_t lag, const mifft_fused_sync *sync, int32_t grid, mifft_stream_t stream);

/*
 * XCD-cooperative form of the same two-pass axis for N = 1024 * 1024, fp32 (csrc/fft_xcd2.hpp): ONE persistent launch of
 * 2 work-groups per CU in which the 64 work-groups resident on each XCD (chiplet) own one transform at a time and
 * hand the inter-pass intermediate over through that XCD's own L2, so that every point crosses the L2 <-> fabric
 * boundary once in and once out (the fused / chained forms: twice).  Same pass pair as mifft_launch_fused2.
 *   scratch  caller-owned, MIFFT_XCD2_SCRATCH_BYTES, 256-byte aligned; contents are don't-care
 *   control  caller-owned, MIFFT_XCD2_CONTROL_BYTES (zeroed by this call on `stream`).  After completion
 *            ((uint32_t*)control)[1] != 0 means the results are INVALID: bit 0 = a bounded wait timed out, bit 1 = the
 *            launch did not find exactly 64 resident work-groups per XCD (nothing was written; run another strategy).
 *   flags    bit 0: issue the next transform's loads while the current one is being stored (default form)
 *            bit 1 (development): `control` is MIFFT_XCD2_CONTROL_BYTES + MIFFT_XCD2_TRACE_BYTES long and receives, behind
 *            the control words, 32 time stamps (100 MHz) per work-group for the per-XCD transform index (flags >> 8)
 *            bits 4..6 (development): elimination variant 1..5 of csrc/fft_xcd2.hpp -- same launch with a part of the work
 *            removed; the results are WRONG by construction (profiles/r03_xcd2_elimination.log)
 *            bits 8..23 the traced transform index; bits 24..30 (development, round 5): the XCDs with an odd id start that many
 *            microseconds late (anti-phase HBM bursts, profiles/r05_xcd2_antiphase.log); results unchanged
 * Requires a device with 8 XCDs x 32 CUs (MI355X); MIFFT_E_UNSUPPORTED otherwise or for other lengths.
 */
#define MIFFT_XCD2_SCRATCH_BYTES (8u * 64u * 16u * 256u * 8u)
#define MIFFT_XCD2_CONTROL_BYTES ((64u + 2u * 512u * 32u) * 4u) /* one 128-byte line per flag */
#define MIFFT_XCD2_PREFETCH 1
#define MIFFT_XCD2_TRACE 2
#define MIFFT_XCD2_TRACE_BYTES (512u * 32u * 8u)
int mifft_launch_xcd2(const mifft_pass *p0, const mifft_pass *p1, const void *in0, const void *in1, void *out0,
                      void *out1, void *scratch, void *control, int32_t flags, mifft_stream_t stream);

/*
 * ---- extensions the reference lists as TODO (TODO.txt:6-8): tiles of a bigger array, sizes that are not powers of two ----
 * Both are built from the power-of-two plans plus two streaming helpers (pyfft_amd/generic.py):
 *   mifft_aux_copy      general strided complex copy over an index space of up to 6 dimensions (dims[0] fastest): gathers the
 *                       tiles of a parent array, or the lines of one axis, into dense rows and scatters them back.  Element
 *                       = one complex number of `precision`; either side may be two scalar planes.  Optional, in this order:
 *                       zero fill for dims[0] indices >= src_valid0 (0 = no padding), conjugation of the input, a complex
 *                       multiplier mult[i0] (device table of dims[0] entries; the chirp of Bluestein's algorithm), a real
 *                       scale, conjugation of the output.  Strides are in elements.
 *   mifft_aux_mul_rows  a[r][j] *= b[j] for `rows` dense rows of n complex numbers (interleaved).
 */
/*
 * Tiled batches in ONE launch (csrc/fft_nd2t.hpp): a MIFFT_PASS_ND pass (L, M, S = the TILE's x, y, z; interleaved) whose
 * `outer` transforms are the non-overlapping tiles of `outer / (cx*cy*cz)` parent arrays, taken from and written to their places
 * in the parent -- tile g = item*(cx*cy*cz) + (iz*cy + iy)*cx + ix starts at item*parent_elems + iz*z*pitch_z + iy*y*pitch_y + ix*x.
 * In place or out of place (same geometry on both sides).  Exists for a list of tile shapes only:
 *   mifft_nd_tiled_supported  0 if the tile shape has such a kernel, else MIFFT_E_UNSUPPORTED
 * mifft_launch_nd_tiled_split is the same launch on split-complex parents (pass->layout = MIFFT_SPLIT; re / im planes with the
 * same element offsets on both sides), for the same list of tile shapes.
 */
typedef struct mifft_tiling {
    int64_t pitch_y;       /* elements between consecutive y of the parent array (its x extent) */
    int64_t pitch_z;       /* elements between consecutive z of the parent array (x extent * y extent) */
    int64_t parent_elems;  /* elements per parent array */
    int32_t cx, cy, cz;    /* tiles per parent axis */
} mifft_tiling;
int mifft_nd_tiled_supported(int32_t precision, int32_t x, int32_t y, int32_t z);
int mifft_launch_nd_tiled(const mifft_pass *pass, const mifft_tiling *tiling, const void *in, void *out, mifft_stream_t stream);
int mifft_launch_nd_tiled_split(const mifft_pass *pass, const mifft_tiling *tiling, const void *in_re, const void *in_im, void *out_re,
                                void *out_im, mifft_stream_t stream);

typedef struct mifft_copy {
    int32_t precision;          /* MIFFT_F32 | MIFFT_F64 */
    int32_t ndim;               /* 1 .. 6 */
    int64_t dims[6];
    int64_t src_stride[6];
    int64_t dst_stride[6];
    int64_t src_valid0;         /* source extent along dims[0]; larger indices read as zero (0: all valid) */
    int32_t src_split;          /* source is two scalar planes (src0 = re, src1 = im) */
    int32_t dst_split;
    int32_t conj_in;
    int32_t conj_out;
    const void *mult;           /* device: dims[0] complex multipliers, or NULL */
    double  scale;
} mifft_copy;
int mifft_aux_copy(const mifft_copy *copy, const void *src0, const void *src1, void *dst0, void *dst1, mifft_stream_t stream);
int mifft_aux_mul_rows(int32_t precision, void *a, const void *b, int64_t rows, int64_t n, mifft_stream_t stream);
/* Verification helper (round 6; no counterpart in the reference, whose tests copy everything to the host: test/test_errors.py:66-114):
 * ADDS to *count the number of 16-byte words in which a[0 .. nbytes) and b[0 .. nbytes) differ.  a, b 16-byte aligned device buffers,
 * nbytes a multiple of 16; count = any 8-byte aligned uint64 the device can add to atomically (device memory, or pinned host memory from
 * mifft_host_alloc, which the host then reads after synchronising `stream`).  Lets a caller check a periodic data set's result -- every
 * transform bit-identical to its period-mate -- over a buffer far too large to copy back. */
int mifft_aux_count_mismatch(const void *a, const void *b, size_t nbytes, uint64_t *count, mifft_stream_t stream);

/*
 * Mixed-radix rows (csrc/fft_mixed.hip; the reference's TODO.txt:8): `rows` contiguous transforms of SMOOTH length
 * n = 2^a 3^b 5^c 7^d, 2 <= n <= 4096 (fp32) / 2048 (fp64), interleaved, row r at element r * stride on either side, in place or
 * out of place; tw = device table of n entries w(n)^m.  out = scale * DFT(in) (inverse: conjugated in and out).
 *   mifft_mixed_supported  0 if rows of n points have such a kernel, else MIFFT_E_UNSUPPORTED
 */
int mifft_mixed_supported(int32_t precision, int32_t n);
int mifft_launch_mixed_rows(int32_t precision, int32_t n, int64_t rows, int64_t stride_in, int64_t stride_out, const void *in,
                            void *out, const void *tw, int32_t inverse, double scale, mifft_stream_t stream);
/* The same transform along ANY axis of a dense array viewed as [outer][n][inner] (inner = product of the faster axes; inner == 1
 * is the row form): line (o, j) starts at o*n*inner + j, its points are `inner` apart; a work-group takes adjacent lines, so the
 * accesses coalesce across lines.  conj_in / conj_out conjugate on load / store separately (an N-D inverse conjugates once at
 * either end of its chain of launches).  In place or out of place. */
int mifft_launch_mixed_lines(int32_t precision, int32_t n, int64_t outer, int64_t inner, const void *in, void *out, const void *tw,
                             int32_t conj_in, int32_t conj_out, double scale, mifft_stream_t stream);
/* LONG smooth lengths, N = n1 * n2 beyond one tile, in two launches (four-step form without a separate transposition): lines of n1
 * points n2 apart, stored as rows and multiplied by w(N)^(k1 * i2) into `mid`; then lines of n2 points n1 apart from `mid` to `out`
 * in place order = natural order.  `mid` must not be `in`; it may be `out` (out-of-place transforms need no scratch).
 *   mifft_mixed_long_split   0 and the factors (both with a mixed-radix kernel, as many adjacent lines per tile as possible) or
 *                            MIFFT_E_UNSUPPORTED (N not smooth, N > 2^24, or no split with >= 8 (fp64: 4) lines per tile)
 *   tables: tw1 / tw2 = w(n1)^m / w(n2)^m; w(N)^e = tw_lo[e & (2^tw_shift - 1)] * tw_hi[e >> tw_shift], e < N */
int mifft_mixed_long_split(int32_t precision, int64_t n, int32_t *n1, int32_t *n2);
int mifft_launch_mixed_long(int32_t precision, int32_t n1, int32_t n2, int64_t batch, const void *in, void *mid, void *out,
                            const void *tw1, const void *tw2, const void *tw_lo, const void *tw_hi, int32_t tw_shift,
                            int32_t inverse, double scale, mifft_stream_t stream);
/* Bluestein's algorithm in ONE launch for rows of ANY length n whose padded length fits the LDS of a CU twice (2 n - 1 <= 10000 fp32 /
 * 5000 fp64; up to 4096 / 2048 several rows share a work-group): both m-point transforms of the convolution run inside LDS
 * (csrc/fft_mixed.hip).
 *   mifft_bluestein_padded   0 and the padded length m (smooth, >= 2 n - 1, the cheapest one) or MIFFT_E_UNSUPPORTED
 *   tables: tw = w(m)^j (m entries); chirp c[j] = exp(-i pi j^2 / n) (n entries); bhat = FFT_m(b) / m with b[j] = conj(c[j]) for
 *   j < n, b[m - j] = b[j], zero between (m entries).  out = scale * DFT(in) (inverse: conjugated in and out). */
int mifft_bluestein_padded(int32_t precision, int32_t n, int32_t *m);
int mifft_launch_bluestein_rows(int32_t precision, int32_t n, int32_t m, int64_t rows, int64_t stride_in, int64_t stride_out,
                                const void *in, void *out, const void *tw, const void *chirp, const void *bhat, int32_t inverse,
                                double scale, mifft_stream_t stream);

/* Whole SMOOTH 2-D / 3-D transforms in one launch (csrc/fft_mixed_nd.hip): every axis of the (z, y, x) shape a smooth length (or 1), at
 * least two axes longer than 1, x * y * z <= 16384 points (fp32) / 8192 (fp64) -- the transform, or several, lives in one work-group's
 * LDS (one buffer; a stage holds its operands in registers across a barrier) between its first load and its last store.  `transforms` dense arrays one after the other, interleaved, in place or out of place;
 * tw_x / tw_y / tw_z = device tables w(len)^m of the axis lengths (NULL for an axis of length 1).  out = scale * DFT3(in);
 * inverse: 0 forward, 1 inverse (input and output conjugated), 2 / 4 conjugate the input / the output only (the first / last launch
 * of a composition, e.g. the (y, x) planes of a bigger 3-D shape followed by mifft_launch_mixed_lines along z).
 *   mifft_mixed_nd_supported  0 if the shape has this form, else MIFFT_E_UNSUPPORTED */
int mifft_mixed_nd_supported(int32_t precision, int32_t x, int32_t y, int32_t z);
int mifft_launch_mixed_nd(int32_t precision, int32_t x, int32_t y, int32_t z, int64_t transforms, const void *in, void *out,
                          const void *tw_x, const void *tw_y, const void *tw_z, int32_t inverse, double scale, mifft_stream_t stream);

/* Same as mifft_launch_chain but brackets the chain with two events on `stream` and, after
 * synchronising, reports the elapsed device time of `repeats` back-to-back chains. (bench/test helper) */
int mifft_time_chain(const mifft_pass *passes, int32_t npasses, void *const bufs0[3], void *const bufs1[3],
                     mifft_stream_t stream, int32_t repeats, float *ms_total);

#ifdef __cplusplus
}
#endif
#endif /* MIFFT_H */
