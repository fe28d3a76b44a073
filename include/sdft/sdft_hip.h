/*
 * sdft/sdft_hip.h -- additions of the MI355X engine beyond the reference's C API.  Nothing here
 * alters the drop-in surface of sdft/sdft.h; a host that ignores this file behaves like one
 * built against the reference.  Included by sdft/sdft.h (needs its type selection).
 */

#ifndef SDFT_HIP_SDFT_HIP_H
#define SDFT_HIP_SDFT_HIP_H

#ifndef SDFT_HIP_SDFT_H
#error "include <sdft/sdft.h>, not <sdft/sdft_hip.h>"
#endif

#if defined(__cplusplus)
extern "C" {
#endif

/* ---- error channel -------------------------------------------------------------------------
   The reference has none (void returns, unchecked malloc; sdft.h:413-687).  Signatures are kept;
   on a HIP failure sdft_alloc* returns NULL, other calls leave their outputs untouched, and the
   text of the first failing HIP call is recorded per thread.  Never aborts. */
const char* sdft_hip_last_error(void);       /* NULL if nothing is recorded */
void        sdft_hip_clear_error(void);
/* A call that SUCCEEDED but has something to tell -- a poll loop of the exact-carry kernels timed out and the call was
   re-run with the serial carry pass (get_option "ring_recoveries" counts them) -- leaves a warning, never an error:
   its outputs and the stream state are valid and the host must not repeat it. */
const char* sdft_hip_last_warning(void);     /* NULL if nothing is recorded */
void        sdft_hip_clear_warning(void);

/* ---- device selection (one process per GPU is the intended multi-GPU model) ----------------- */
int         sdft_hip_device_count(void);
int         sdft_hip_set_device(int device); /* plans are created on the current device */
int         sdft_hip_get_device(void);
const char* sdft_hip_version(void);
int         sdft_hip_selftest(void);         /* 0 = cross-lane primitives behave as the kernels assume */
/* compiles the statements of a sdft_hip_op_expr operation (below) for `arch` (NULL: "gfx950") without running them:
   0, or -1 with the compiler's words in sdft_hip_last_error().  Needs no GPU. */
int         sdft_hip_check_expr(const char* expr, const char* arch);

/* measurement aid: average ms of a store-only kernel over `bytes` of device memory.
   pattern 0 = linear fill; pattern 1 = the forward kernel's tiling (rows of `row_slots` 16-byte
   slots, `lanes` slots per wave, `chunk_len` consecutive rows per wave); pattern 2 = one workgroup per time chunk writing
   whole rows in lockstep (`lanes` = rows per barrier); pattern 3 = the same with non-temporal stores; 4 = every XCD a contiguous
   eighth of the chunks (what ForwardArgs::xcd_map does in the analysis kernels); 5 = workgroups started at different row phases;
   6 = both; 100 + R = workgroup b takes the (b / R)-th chunk of region b mod R */
double      sdft_hip_store_ceiling(void* dst, size_t bytes, int pattern, unsigned row_slots, unsigned lanes,
                                   unsigned chunk_len, int reps);
/* Device memory for a DFT matrix, chosen for how fast it can be written: which physical memory backs a large allocation decides what
   the analysis' store stream reaches in it (5.85, 6.4 or 7.1 TB/s for 16 GB buffers of one process; reads do not care) and nothing
   can move a buffer afterwards.  Allocates up to `candidates` buffers of `bytes` (as many as fit beside each other), probes each with
   the store-only kernel above (2 launches), keeps the best and frees the others; *gbs (may be NULL) = the kept buffer's probe rate.
   Free with hipFree.  NULL on failure. */
void*       sdft_hip_malloc_matrix(size_t bytes, int candidates, double* gbs);
/* The same choice made inside ONE allocation, computed rather than searched (round 6, profiles/r06_stretch_map.txt): a large allocation is
   made of stretches of two or three KINDS of memory that alternate every 16-32 GiB (the first change lies 32 or 64 GiB into the allocation in
   every session kept), and the analysis' store stream -- every XCD writing its own eighth of the matrix at the same time -- reaches 6.8-7.1 TB/s
   when the matrix lies half in one kind and half in another, 5.6-5.85 when all of it is of one kind.  Allocates `arena_bytes` (>= bytes;
   bytes + 64 GiB has held a change of kind in every fresh process so far), finds the first change with small two-part store probes (2 GiB written each, steps
   of 16 GiB then bisection: <= 12 ms in all), returns the window centred on it and keeps the whole allocation until
   sdft_hip_free_matrix(window) -- NOT hipFree: the window is not the start of the allocation.  Two full-size probes check the result (the
   window, and a window at the allocation's start = what a plain hipMalloc would have been) and the better one is returned.  An allocation that
   holds no change of kind at all (seen in a process that had allocated and freed a lot before) is answered by a second allocation made while the first
   is held -- other memory by construction -- and the better of the two is kept; an arena too small for the two-part probes (< 3 GiB) is searched window
   by window (every 4 GiB, at most 8).  *gbs (may be NULL) = the window's probe rate.  NULL on failure; free: 0, or -1 for a pointer this call did
   not return.  Matrices below 64 MiB are not probed and get an allocation of their own size.  sdft_hip_matrix_placement tells how a window was placed. */
typedef struct
{
  size_t arena_bytes;      /* the allocation that holds the window */
  size_t window_offset;    /* of the window inside it */
  size_t boundary_offset;  /* where the kind of memory changes (0: no change found, the window came from the search or is the start) */
  int    pair_probes;      /* two-part probes of 2 GiB used to find it */
  int    window_probes;    /* full-size store-only probes (the start, the window; more only when the search ran) */
  double window_gbs;       /* store-only rate of the window, GB/s */
  double start_gbs;        /* ... of a window at the allocation's start */
  double probe_ms;         /* GPU time of all probes */
  int    arenas_tried;     /* 1; 2 when the first allocation held no change of kind and a second one was made beside it (the better is kept) */
} sdft_hip_placement_t;
void*       sdft_hip_malloc_matrix_in_arena(size_t bytes, size_t arena_bytes, double* gbs);
int         sdft_hip_matrix_placement(const void* window, sdft_hip_placement_t* out);
int         sdft_hip_free_matrix(void* window);
/* measurement aid: occupies `cus` CUs (nothing shares them) for `milliseconds` on a stream of its own and returns at once;
   cus = 0 waits for the release.  What a kernel keeps of its speed beside a kernel that holds part of the chip. */
int         sdft_hip_hold_cus(unsigned cus, double milliseconds);
/* the same for a load-only kernel (16-byte loads, four in flight per thread) */
double      sdft_hip_load_ceiling(const void* src, size_t bytes, int reps);
/* ... and for whole rows of `row_slots` 16-byte slots read in step by one workgroup per chunk of `chunk_len` rows
   (regions = 0: workgroup b -> chunk b; R: regions in turn, as store pattern 100 + R) */
double      sdft_hip_load_rows_ceiling(const void* src, size_t bytes, unsigned row_slots, unsigned chunk_len, unsigned regions, int reps);

/* ---- batched plans: `channels` independent streams with one launch per call ----------------
   The unit of sharding in the reference is the plan (no shared mutable state, sdft.h:145-182);
   a batch is the same thing laid out for one GPU.  With a batched plan
     sdft_sdft_n : samples [channels][nsamples],  dfts [channels][nsamples][dftsize]
     sdft_isdft_n: dfts as above,                 samples [channels][nsamples]          */
sdft_t*     sdft_hip_alloc_batch(const sdft_size_t dftsize, const sdft_window_t window,
                                 const sdft_double_t latency, const sdft_size_t channels) SDFT_HIP_SYMBOL(alloc_batch);
sdft_size_t sdft_hip_channels(const sdft_t* sdft) SDFT_HIP_SYMBOL(channels);

/* ---- fused analysis -> spectral operation -> synthesis -------------------------------------------
   The reference's contract materialises the (nsamples, dftsize) matrix between sdft_sdft_n and
   sdft_isdft_n (README.md:42-47 of the reference); a host that only wants the processed signal back
   pays two HBM streams of 16 KiB per sample for it.  sdft_hip_process_n computes
       out[t] = sdft_isdft( op( sdft_sdft(samples[t]) ) )
   with the same arithmetic and the same plan state update as the two calls, but the rows stay
   inside the workgroup that produced them (dftsize 8 ... 4096; other shapes, and the reference's summation
   order beyond 2048 double / 4096 float bins, run analysis + synthesis back to back through a workspace).
     sdft_hip_op_identity  params = NULL
     sdft_hip_op_gain      params = sdft_fd_t gains[dftsize]  (host or device): X'[k] = X[k] * gains[k]
     sdft_hip_op_shift     params = const long* (host): X'[k] = X[k - *params], zero outside the spectrum
     sdft_hip_op_cgain     params = sdft_fdx_t gains[dftsize] (host or device): X'[k] = X[k] * gains[k], complex
     sdft_hip_op_gain_rows / sdft_hip_op_cgain_rows   gains that change with time -- what a host of the reference does when
                           its loop over the matrix recomputes the mask every hop: params = const sdft_hip_gain_rows_t*
                           { gains[rows][dftsize] (real / complex; host or device), rows, hop }: vector r applies to the
                           call's samples [r*hop, (r+1)*hop), the last vector to everything after it.  Still one launch:
                           the folded coefficients of all vectors are formed by one small launch in front of it.
     sdft_hip_op_gate      params = sdft_fd_t[2] {threshold, floor} (host): X'[k] = X[k] if |X[k]| >= threshold, else
                           X[k] * floor (a spectral gate; floor = 0 removes the bin)
     sdft_hip_op_power     params = sdft_fd_t[2] {exponent, scale} (host): X'[k] = X[k] * scale * |X[k]|^(exponent-1), i.e.
                           |X'[k]| = scale * |X[k]|^exponent with the phase kept (spectral compression / expansion)
                           (gate and power are not linear in the spectrum: they run on the windowed rows inside the
                           row-group kernel, dftsize <= 2048 double / 4096 float, two passes beyond)
     sdft_hip_op_expr      the host's own operation, handed in as code: params = const sdft_hip_expr_t*
                           { expr, params, nparams }.  expr is a string of HIP C++ statements applied to every windowed
                           bin; in scope are  re, im  (sdft_fd_t; read and assign them: the value of bin k),  k, nbins
                           (unsigned),  t  (size_t: index of the sample within the call),  ch  (size_t: channel),
                           p[i]  (sdft_fd_t, read-only: the call's nparams parameters, copied from host memory with the call)
                           and HIP's math functions.  Example (a gate with a frequency-dependent threshold):
                               "if (re * re + im * im < p[0] * p[0] * (1 + k)) { re = 0; im = 0; }"
                           The statements are compiled into the fused kernel at run time (hiprtc: libhiprtc.so is opened
                           on first use; about a second per new expression and kernel shape, then cached for the life of
                           the process), so the call still moves no matrix; a hop (one time chunk) runs the hop kernel and
                           the row synthesis with the statements built in (two launches), rows beyond the row-group
                           kernel run analysis -> expression on the rows -> synthesis.
                           What sdft.h leaves to the host between sdft_sdft_n and sdft_isdft_n (README.md:42-47), on the GPU.
   dfts: NULL, or device memory of shape (nsamples, dftsize) that receives the processed spectrum
   (not with the shift).  Batched plans: samples / out [channels][nsamples].
   Results equal sdft_sdft_n + operation + sdft_isdft_n of the reference within the path's bar (1e-6
   relative at FD double, 1e-4 at FD float; measured 1e-13 / 1.2e-5); bit-identical on request (option
   "fused_exact" = 1) wherever the analysis is: calls shorter than 512 samples, FD float, FD double with
   carry = 1.  The stream state a call leaves behind is the one the two calls leave.  Returns 0, or -1
   with sdft_hip_last_error() set. */
enum sdft_hip_op { sdft_hip_op_identity = 0, sdft_hip_op_gain = 1, sdft_hip_op_shift = 2, sdft_hip_op_cgain = 3,
                   sdft_hip_op_gain_rows = 4, sdft_hip_op_cgain_rows = 5, sdft_hip_op_gate = 6, sdft_hip_op_power = 7,
                   sdft_hip_op_expr = 8 };
typedef struct { const void* gains; size_t rows; size_t hop; } sdft_hip_gain_rows_t;
typedef struct { const char* expr; const void* params; size_t nparams; } sdft_hip_expr_t;
int sdft_hip_process_n(sdft_t* sdft, const sdft_size_t nsamples, const sdft_td_t* samples, sdft_td_t* const out,
                       const int op, const void* params, sdft_fdx_t* dfts) SDFT_HIP_SYMBOL(process_n);

/* ---- streams ---------------------------------------------------------------------------------
   Every plan owns a HIP stream.  Calls with host pointers always return with the output
   complete.  Calls with device pointers do too unless option "async" is 1; then they return after
   enqueueing and sdft_hip_synchronize() (or the caller's own stream sync) completes them.
   Asynchronous analysis calls on the plan's OWN stream may run on internal streams beside it (option "pipeline"):
   sdft_hip_synchronize() and every later call of the plan wait for them, and so does the null stream (they are blocking
   streams like the plan's own: a plain hipMemcpy of the results sees them complete); a host that wants to queue its own work behind
   a call asks for the stream with sdft_hip_get_stream() -- from then on every kernel of the plan is on that stream -- or
   hands the plan a stream of its own with sdft_hip_set_stream(). */
int   sdft_hip_set_stream(sdft_t* sdft, void* hip_stream /* hipStream_t */) SDFT_HIP_SYMBOL(set_stream);
void* sdft_hip_get_stream(sdft_t* sdft) SDFT_HIP_SYMBOL(get_stream);
int   sdft_hip_synchronize(sdft_t* sdft) SDFT_HIP_SYMBOL(synchronize);
/* measurement aid: the reference driver's loop (test/test.c:69-83: sdft_sdft_n + sdft_isdft_n per hop of `hop` samples on one
   matrix) run `hops` times from C, as a C host runs it; seconds, or -1 */
double sdft_hip_time_hops(sdft_t* sdft, size_t hops, size_t hop, const sdft_td_t* samples, sdft_fdx_t* dfts, sdft_td_t* out) SDFT_HIP_SYMBOL(time_hops);

/* ---- options -----------------------------------------------------------------------------------
   Twenty-two keys.  The first twelve are for hosts; the defaults are the safe and (but for the host-memory choices, which only the host can
   make) the fast ones:
   "async"         0|1   see above
   "pipeline"      asynchronous analysis calls on the plan's own stream into matrices that do not overlap (a host that alternates between two
                       matrices) overlap: the state after a call comes from a small kernel ahead of the call's rows, the rows of consecutive
                       calls run on two internal streams; every other call of the plan and sdft_hip_synchronize wait for them.
                       1 (default) = calls of less than 2^29 bins, where the next call fills what a launch leaves idle -- a fixed 20-30 us per call:
                       n = 48 000, m = 1024: 82 against 75 % of the HBM peak, n = 131 072: 85 against 76 % (profiles/r06_pipelined_calls.txt); longer calls
                       amortise that by themselves and stay on one stream (n = 1e6: a tie at 85 %).  2 = calls of any
                       length, 0 = never.  Asynchronous synthesis calls that come back to back take the two streams in turn as well; a synthesis
                       never runs beside an analysis; either kind only once two of them have come in a row.  Off by itself on a caller's stream,
                       once sdft_hip_get_stream has been called, and with profiling.  get_option "last_pipelined", "pipelined_calls",
                       "pipelined_inverse_calls", "pipelined_ordered", "pipeline_streams" (10 x kind + pairs tried; kind 1 = ordinary
                       streams, 2 = by priority: what the plan falls back to when no ordinary pair runs concurrently, 0 = none found)
   "carry"         0 = chunk-parallel carries (FD double default; <= 1e-11 relative deviation from the
                       serial reference), 1 = exact serial carry pass (bit-identical; FD float always)
   "float_carry_parallel"  0 (default) | 1 = FD float plans take the chunk-parallel carries too: long calls run at
                       twice the speed and land closer to the double-precision result than the reference's float
                       arithmetic does, but NOT within 1e-4 of the float reference (which itself drifts about 2e-4
                       of the largest bin per 262144 samples); the state then differs from the reference's by the same
   "exact_inverse" 1 (default) = synthesis gives the reference's bits (bins of a row added in the reference's order, or --
                       float samples from double bins, up to 500 000
                       rows -- a tree sum whose rounding interval proves the reference's float, rows it cannot prove
                       added in order), 0 = wave-parallel tree sum, unverified
   "host_copy"     0 (default) = copies between the caller's host memory and the device go through pinned 2 MiB pieces of the
                       plan (beyond 64 KiB): the runtime is never handed caller memory to pin.  Its own path for pageable
                       memory pins the pages and remembers the pin by address; a host that frees the buffer, lets the heap
                       shrink and gets the address back later makes the next copy fault the GPU (process gone).  1 = the
                       runtime's path: faster (the reference driver's hop, 1.6 MB out and back: 128 against
                       197 us; long copies 55 against 26 GB/s) and safe for a host that allocates its buffers once and
                       keeps them, like the reference's driver (test/test.c:62-83) -- as is "host_register" = 1 (110 us);
                       get_option "host_copies_staged" counts the copies that went through the pieces
   "host_register" 0 (default) = host buffers are copied through staging buffers; 1 = host buffers of 1 MiB and more are
                       registered in place once (hipHostRegister, the last 8 page ranges are remembered) and the kernels
                       read and write them over PCIe: 139 -> 111 us per 100-sample hop of the reference's test driver.
                       ONLY for hosts that keep their buffers allocated while the plan lives (like test/test.c:62-64 of the
                       reference): a registration does not survive free() + malloc() handing the same address out again.
   "copy_threads"  2 (default) = worker threads of the copies between the caller's host memory and the plan's pinned slots (the host's
                       copy of one piece overlaps the DMA of the next; a hop-sized matrix is copied by the workers and the caller
                       together); 0 = the calling thread alone
   "pinned_io"     1 (default) = host sample buffers of up to 64 KiB (a hop of a host signal, the sample of sdft_sdft, the
                       result of sdft_isdft) travel through a pinned scratch of the plan that the kernels access directly
   "spin"          1 (default) = synchronous calls never sleep on the stream while they can still be running: calls of one
                       time chunk poll a completion word their kernel sets in pinned host memory (it is visible ~6 us
                       before the stream reports the kernel finished); other calls spin on the host clock until the call's
                       bytes could have moved at the chip's peak rate, then poll the stream (a sleeping wait wakes up 9 us
                       late on one box and 45 us late on another); 0 = sleep on the stream, 2 = poll from the start
   "profile"       0 off, 1 = HIP events around every stage, 2 = around the forward / inverse kernel only
                       (read with sdft_hip_get_profile; every event pair costs ~5 us of stream time)
   "resident"      0 (default) | 1 = the reference driver's loop -- sdft_sdft_n + sdft_isdft_n per hop, synchronous calls of one time chunk
                       (< 512 samples; synthesis of up to 1024 rows) on device pointers, single-channel plan on its own stream
                       (test/test.c:69-83 of the reference), and the single-sample calls sdft_sdft / sdft_isdft on a device row (the sample rides in
                       the call, the result comes back through pinned memory: 6.8 / 5.8 us per call against 11.8 / 9.9) -- is served by ONE kernel that stays on the chip: the host writes a call into a
                       cache line of pinned memory and rings a doorbell word, the kernel's workgroups run the same device functions the launches
                       run (bit-identical) and set the completion word: no launch and no stream query per call.  The kernel leaves by itself when
                       no call has come for 200 us (the first call after that starts it again), so a blocking hipMemcpy of the host -- which
                       waits for the plan's stream -- waits that long at most; every other entry point of the plan and sdft_hip_synchronize
                       retire it first.  Off by default: while it lives it holds 256 small workgroups and the plan's stream.
                       get_option "resident_calls", "resident_launches", "resident_missed" (calls that raced with the time-out and were
                       rung again), "resident_alive"
   Ten more pick a route the library otherwise picks by itself (a host that knows its shapes may, the tests do):
   "chunk"         samples per time chunk (0 = heuristic)
   "segments"      time segments of the exact carry pass overlapped with the forward launches (0 = heuristic)
   "chain"         exact carries: 1 (default) = relay form (seed table; identical waves take blocks of steps in turn, one dependent addition
                       per step on the chain; one relay launch beside one forward launch whose workgroups wait for their chunk's carries)
                       while bins x channels leave SIMDs idle, 0 = always the serial pass, 2 = relay form whenever the geometry allows
   "self_carry"    1 (default) = chunk-parallel FD double calls with 2*dftsize <= 4096 a power of two or 2/3/5-smooth run as ONE
                       launch: every workgroup derives its carry-in from the raw samples (fold + FFT in LDS);
                       0 = carries by a pre-pass (two more launches); calls of up to 2^19 samples per channel take it
   "hop_kernel"    1 (default) = calls of one time chunk run one fused launch (differences + analysis; a hop's samples in up to 8 time
                       parts, every (tile of bins, part) a workgroup: bit-identical), 0 = the general launches
   "fused_exact"   sdft_hip_process_n: 0 = folded form (window, operation and synthesis folded into per-bin
                       coefficients, sum over bins by a tree), 1 = bins summed in the reference's order by the
                       fastest route that gives those bits, 2 = in that order by the fused kernel,
                       -1 (default) = 1 when the host set carry = 1 at FD double, else 0
                       (float samples from double bins: the reference's bits are proven from the tree sum and a bound on
                       what any summation order can differ by; only samples whose bound straddles a rounding boundary of the
                       float are summed in order -- get_option "ordered_walks" counts them)
   "inverse_rows"  rows per wave of the exact inverse (0 = heuristic and tuner; 4, 8, 16, 32)
   "inverse_tune"  1 (default) = synthesis calls from 8 Ki rows on find the fastest of their bit-identical forms -- 4, 8, 16 or 32 rows
                       per wave, 256- or 512-byte row segments, the tree sum with the rounding-interval proof, whole rows read in step -- on the
                       host's own calls: the first calls of a shape take the forms in turn, timed by events, then the fastest serves the shape
                       (a few shapes are remembered: a host that alternates call lengths keeps what it has decided); 0 = the static choice.
                       get_option "last_inverse_tuned" = 10 x decided + form (0 tree sum, 1 32 rows, 2 16 rows, 3 16 rows x 512 B, 4 8 rows,
                       5 4 rows, 6 whole rows in step)
   "pointers"      0 (default) = every call asks the runtime what each pointer is (hipPointerGetAttributes: 0.06-0.16 us,
                       nothing is cached -- a buffer that was freed and whose address came back as the other kind of memory
                       is classified as what it is now), 1 = all device, 2 = all host (no query)
   "stage_bytes"   segment size of the host-pointer staging path
   A key the library does not know returns -1.
   TEST HOOKS.  Every other fork of the host logic is decided by the library alone in libsdft_hip.so.  The same sources built with
   -DSDFT_HIP_TEST_HOOKS (libsdft_hip_hooks.so, built beside the product by `python -m sdft_amd.build`; no host links it) accept the keys that force
   those forks, so that the tests can run every route against the reference and the probes under scripts/ can measure them:
   "rows_kernel", "row_slots_max", "interior", "fused", "fft_carry", "fold", "rows_f32", "hop_parts", "xcd_map", "chain_block", "relay_waves", "inverse_verify",
   "relay_flow", "relay_groups", "chain_debug", "inverse_nt", "inverse_nt_skip_mb", "inverse_step", "inverse_ordered", "host_direct", "copy_streams" (sdft_capi.inc names what each selects);
   get_option "test_hooks" = 1 in that build.
   get_option additionally answers "tiles", "bins_per_lane", "row_slots", "last_chunks",
   "last_chunk_len", "last_kernel" (1 tiles, 2 row groups, 3 hop), "last_segments", "last_fused",
   "last_chain", "last_fused_exact", "last_fused_fold", "last_process_path" (1 fused kernel, 2 hop pair, 3 two-pass),
   "last_self", "last_inverse_nt" / "last_inverse_skip" (what the last synthesis launch used: non-temporal loads, rows read with ordinary loads),
   "cursor", "device", "ring_recoveries" (calls re-run with the serial carry pass after a poll loop of
   the exact-carry kernels timed out: results stay valid, sdft_hip_last_warning() reports it), "flag_fallbacks". */
int  sdft_hip_set_option(sdft_t* sdft, const char* key, long value) SDFT_HIP_SYMBOL(set_option);
long sdft_hip_get_option(const sdft_t* sdft, const char* key) SDFT_HIP_SYMBOL(get_option);

/* accumulated device milliseconds and launch counts per stage {delta, carry, forward, inverse};
   synchronises, then resets the counters */
int  sdft_hip_get_profile(sdft_t* sdft, double ms[4], long calls[4]) SDFT_HIP_SYMBOL(get_profile);

/* ---- introspection (tests) ---------------------------------------------------------------------
   stream state copied to host buffers: acc, fid [channels][dftsize]; hist [channels][2*dftsize]
   in time order (oldest first); any pointer may be NULL */
int  sdft_hip_get_state(sdft_t* sdft, sdft_fdx_t* acc, sdft_fdx_t* fid, sdft_td_t* hist, size_t* cursor) SDFT_HIP_SYMBOL(get_state);

/* checkpoint / resume: installs a state read with sdft_hip_get_state into a plan of the same dftsize,
   window, latency, types and channel count (also one living on another GPU); NULL pointers leave
   that part untouched */
int  sdft_hip_set_state(sdft_t* sdft, const sdft_fdx_t* acc, const sdft_fdx_t* fid, const sdft_td_t* hist, size_t cursor) SDFT_HIP_SYMBOL(set_state);

/* host-only (no GPU needed): the plan tables exactly as uploaded; tw, syn [dftsize], wtab
   [2*dftsize], weights [2] = {analysis, synthesis}; any pointer may be NULL */
int  sdft_hip_plan_tables(const sdft_size_t dftsize, const sdft_double_t latency, sdft_fdx_t* tw,
                          sdft_fdx_t* syn, sdft_fdx_t* wtab, sdft_fd_t* weights) SDFT_HIP_SYMBOL(plan_tables);

#if defined(__cplusplus)
}
#endif

#endif /* SDFT_HIP_SDFT_HIP_H */
