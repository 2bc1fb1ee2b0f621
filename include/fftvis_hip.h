/* fftvis_hip.h -- C ABI of libfftvis_hip.so, the MI355X (gfx950) backend that fills the
 * reference's stubbed `gpu` backend (tyler-a-cox/fftvis, src/fftvis/gpu/).
 *
 * Plain C, plain pointers and sizes; no C++/torch types cross this boundary.  Every entry point
 * returns an int status (0 = OK, see FV_* below); no exception crosses the ABI.  After a
 * non-zero status fv_last_error() returns a message owned by the library, valid until the next
 * failing call on the same host thread.  Host arrays are caller-owned, contiguous, never
 * modified; complex data is interleaved (re, im).  "precision" is the reference's:
 * 1 = float32/complex64, 2 = float64/complex128 (src/fftvis/cpu/cpu_simulate.py:591-596).
 *
 * Each declaration cites the reference interface it stands behind.
 */
#ifndef FFTVIS_HIP_H
#define FFTVIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV_OK 0
#define FV_ERR_ARG 1
#define FV_ERR_HIP 2
#define FV_ERR_ROCFFT 3 /* reserved (library FFT no longer used) */
#define FV_ERR_INTERNAL 4

/* ---- discovery / diagnostics ------------------------------------------------------------ */
int fv_version(void);                 /* 10000*major + 100*minor + patch */
int fv_device_count(int *count);      /* number of visible HIP devices (0 on a CPU-only box) */
int fv_device_bytes(int64_t *bytes);  /* device memory this process's handles hold right now (all devices) */
/* ... and the share of it on `device`: what the next run there reuses (its cached handle's buffers) -- the term the
 * host side adds to the free memory when it sizes chunks and blocks (wrapper.py:292-302 measures free host RAM).  */
int fv_device_bytes_on(int device, int64_t *bytes);
/* Free / total device memory of `device` (hipMemGetInfo): what the host side sizes its (time, frequency)
 * output blocks and source chunks against -- the device counterpart of the reference's
 * psutil.virtual_memory().available in simulate_vis (src/fftvis/wrapper.py:292-302).                 */
int fv_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes);
/* fv_nufft3 / fv_nudft3_direct keep a stream, device buffers and a plan per host thread between calls
 * (while the process holds < FFTVIS_HIP_HANDLE_CACHE_BYTES, default 2 GiB, of device memory): this frees
 * them all, including those of threads that have exited.  Call it while no transform is in flight. */
int fv_release_workspaces(void);
const char *fv_last_error(void);

/* ---- standalone type-3 NUFFT ---------------------------------------------------------------
 * out[t][k] = sum_j c[t][j] exp(+i (s_k x_j + t_k y_j [+ u_k z_j])),   relative l2 error ~ eps.
 * Replaces: gpu_nufft2d / gpu_nufft3d stubs  (src/fftvis/gpu/nufft.py:11-50, 53-98), i.e. the
 * GPU twins of cpu_nufft2d / cpu_nufft3d -> finufft.nufft2d3 / nufft3d3
 * (src/fftvis/cpu/nufft.py:48-59, 105-118; modeord irrelevant for type 3, isign = +1 default).
 * dim = 2: z and u must be NULL.  x,y,z: (M) reals; c: (ntrans, M) complex; s,t,u: (N) reals;
 * out: (ntrans, N) complex, caller-allocated.  upsampfac in {2.0, 1.25}
 * (cpu/nufft.py:19 "upsample_factor"); 1.25 reaches ~1e-8 at best in fp64 (kernel one cell wider
 * than finufft's formula, capped at 15: beyond that amplified rounding at band-edge targets
 * outweighs the truncation gain).  Host pointers.  NaN / infinite source coordinates are refused
 * (FV_ERR_ARG), as finufft refuses them.                                                      */
int fv_nufft3(int device, int precision, int dim, int64_t M, const void *x, const void *y,
              const void *z, const void *c, int ntrans, int64_t N, const void *s, const void *t,
              const void *u, double eps, double upsampfac, void *out);

/* Same transform by brute force on the GPU (O(M N) direct sum in fp64 accumulators): an
 * independent device-side check used by the parity tests at sizes the CPU oracle cannot reach. */
int fv_nudft3_direct(int device, int precision, int dim, int64_t M, const void *x, const void *y,
                     const void *z, const void *c, int ntrans, int64_t N, const void *s,
                     const void *t, const void *u, void *out);

/* ---- stand-alone pieces of the slice (the other stubbed GPU entry points) -----------------
 * Host pointers; real/complex of `precision`.
 *
 * fv_beam_eval: GPUBeamEvaluator.evaluate_beam (src/fftvis/gpu/beams.py:18-66; CPU twin
 * cpu/beams.py:12-89).  kind/diameter/table as in fv_sim_set_beam_*; out is (2,2,n) complex
 * [ax][feed][src] when polarized, else (n) complex power.  For kind 0 a non-NULL `table` holds 9
 * float64: the four complex Jones factors and the power factor of fv_sim_set_beam_airy_scaled.
 * fv_apparent_coherency: GPUBeamEvaluator.get_apparent_flux_polarized (gpu/beams.py:68-88) and
 * its CPU siblings (cpu/beams.py:129-246, cpu_simulate.py:183-187); variant 0..4:
 *   0 (A^H A) I   1 A^H C A   2 Ai^H Aj I   3 Ai^H C Aj   4 sqrt(Bi Bj) I (1-D arrays).
 *   beam_i/beam_j/out: (2,2,n) complex (variant 4: (n)); flux: (n) real or (2,2,n) complex.
 * fv_inplace_rot: gpu.utils.inplace_rot (src/fftvis/gpu/utils.py:8-22): b (3,n) <- rot (3,3) b. */
int fv_beam_eval(int device, int precision, int polarized, int kind, double diameter,
                 int nfreq_tab, int nza, int naz, double za_max, const void *table, int order,
                 int freq_index, double freq, int64_t n, const void *az, const void *za, void *out);
int fv_apparent_coherency(int device, int precision, int variant, int64_t n, const void *beam_i,
                          const void *beam_j, const void *flux, void *out);
int fv_inplace_rot(int device, int precision, const double *rot, void *b, int64_t n);
/* fv_astrom_topo: one time step of the device-side coordinate manager (see fv_sim_set_astrom): eq (3, n) ICRS unit
 * vectors -> topo (3, n) topocentric (east, north, up) unit vectors under one 31-double context -- what matvis'
 * CoordinateRotationERFA.rotate(t) leaves in all_coords_topo (cpu_simulate.py:937).                      */
int fv_astrom_topo(int device, int precision, const double *astrom, int64_t n, const void *eq, void *topo);

/* ---- fused simulator (the hot loop) ---------------------------------------------------------
 * One handle = one GPU context: streams, FFT twiddle / deconvolution tables, device-resident catalog / baselines /
 * beams / scratch.  A handle is not thread-safe; different handles are independent.
 * Replaces: GPUSimulationEngine._evaluate_vis_chunk stub (src/fftvis/gpu/gpu_simulate.py:62-91),
 * i.e. the GPU twin of CPUSimulationEngine._evaluate_vis_chunk
 * (src/fftvis/cpu/cpu_simulate.py:856-1071) and the helpers it calls
 * (_compute_apparent_coherency :90-202, _run_nufft :205-300, cpu/beams.py:129-246,
 *  cpu/utils.py:5-24).                                                                        */
typedef struct fv_sim fv_sim;

/* precision 1|2; eps: NUFFT accuracy (core/simulate.py:16-19 defaults are the caller's job);
 * upsampfac 2.0|1.25 (cpu/nufft.py:19 "upsample_factor", handed to finufft as is), or 0 = let every
 * fv_sim_run pick: 1.25 when eps >= 1e-8 (fp32: 1e-4; ten times that in 3-D), the fine grid at sigma = 2 has >= 4e6 cells
 * and >= 30 cells per source and target (200 in 3-D), else 2 -- the accuracy contract is eps either way;
 * polarized: nfeeds = 2 (cpu_simulate.py:589). */
int fv_sim_create(fv_sim **h, int device, int precision, double eps, double upsampfac,
                  int polarized);
int fv_sim_destroy(fv_sim *h);

/* Source catalog.  eq: (3, nsrc) equatorial unit vectors (x = cos dec cos ra, ...), real.
 * flux: coherency as prepared by prepare_source_catalog (cpu/utils.py:26-80), already x0.5:
 *   polarized_sky = 0: (nsrc, nfreq) real;  = 1: (nsrc, nfreq, 2, 2) complex.
 * on_device != 0: pointers are device pointers on this handle's GPU (e.g. tensors received by
 * an RCCL broadcast); the library copies either way and the caller keeps ownership.          */
int fv_sim_set_sources(fv_sim *h, int64_t nsrc, int nfreq, const void *eq, const void *flux,
                       int polarized_sky, int on_device);

/* Per-time equatorial -> topocentric ENU rotation matrices, (ntimes, 3, 3) float64 row-major:
 * the device-side stand-in for matvis CoordinateRotation.rotate/select_chunk as used at
 * cpu_simulate.py:937-946 (above-horizon selection up > 0 happens on the device).            */
int fv_sim_set_times(fv_sim *h, int ntimes, const double *rot_eq2enu);

/* Alternative to fv_sim_set_times for callers that own an astrometry engine (matvis
 * CoordinateRotationERFA/Astropy): per-time topocentric ENU unit vectors of EVERY catalog source,
 * (ntimes, 3, nsrc) real of the handle's precision -- what coord_mgr.rotate(ti) produces at
 * cpu_simulate.py:937 before the horizon cut.  Call after fv_sim_set_sources.                 */
int fv_sim_set_topo(fv_sim *h, int ntimes, int64_t nsrc, const void *topo, int on_device);

/* The coordinate manager on the device (the matvis manager the CPU engine builds at
 * cpu_simulate.py:693-709 and rotates per time at :937): instead of per-source vectors the caller hands over the
 * SOURCE-INDEPENDENT context of each time, astrom (ntimes, 31) float64 = one eraASTROM per time in ERFA's field order
 *   pmt, eb[3], eh[3] (Sun -> observer unit vector), em (au), v[3] (observer barycentric velocity / c), bm1,
 *   bpn[3][3] (row-major), along, phi, xpl, ypl, sphi, cphi, diurab, eral, refa, refb
 * -- what erfa.apco13 / astropy's erfa_astrom.apco fills in microseconds -- and the library applies it to every
 * catalog source in front of the horizon cut: light deflection by the Sun, annual aberration, bias-precession-
 * nutation, Earth rotation angle + longitude (eral), polar motion, diurnal aberration, rotation to the horizon,
 * refraction (refa = refb = 0: none) -- the published eraAtciqz / eraAtioq algorithms.  No (ntimes, 3, nsrc)
 * host stream (1.4 GB per 60 times at 1e6 sources).  A context with bpn = identity, v = 0, em huge, xpl = ypl =
 * diurab = refa = refb = 0 and eral = local sidereal angle is exactly fv_sim_set_times' rotation.
 * Replaces fv_sim_set_times / fv_sim_set_topo.  Parity versus ERFA itself is unpinned in this pipeline.    */
int fv_sim_set_astrom(fv_sim *h, int ntimes, const double *astrom);

/* Frequencies (Hz), float64 (nfreq). (cpu_simulate.py:969-973) */
int fv_sim_set_freqs(fv_sim *h, int nfreq, const double *freqs);

/* Array: rotation_matrix (3,3) float64 applied to topo before the NUFFT (cpu_simulate.py:961-962),
 * bls (3, nbls) float64 in SECONDS = R (a2 - a1) / c (cpu_simulate.py:650-659), is_coplanar
 * (cpu_simulate.py:655).                                                                      */
int fv_sim_set_array(fv_sim *h, const double *rotation_matrix, int64_t nbls, const double *bls,
                     int is_coplanar);

/* Lattice ("gridded") array: the type-1 path the reference takes by default for flat arrays whose
 * antennas sit on a lattice (cpu_simulate.py:634-637, 661-681; cpu_nufft2d_type1,
 * cpu/nufft.py:120-175).  basis_matrix (3,3) float64 in SECONDS (= lattice basis / (factor c),
 * :676) -- topo is rotated by its transpose (:964-965); bls_int (2, nbls) int32 lattice
 * coordinates of each baseline (:666-670); n_modes = 2 max|bls_int| + 1 (:673).  Alternative to
 * fv_sim_set_array; results equal the type-3 path's to the NUFFT accuracy.                     */
int fv_sim_set_array_type1(fv_sim *h, const double *basis_matrix, int64_t nbls, const int *bls_int,
                           int n_modes);

/* Beams (evaluate_beam, cpu/beams.py:12-89).  kind 0: analytic Airy dish, param[0] = diameter
 * [m]; E-field 2 J1(x)/x in all four Jones slots, power beam = its square.
 * kind 1: tabulated on a regular (za, az) grid; table is
 *   polarized:   (nfreq_tab, 2, 2, nza, naz) complex128  [ax, feed]
 *   unpolarized: (nfreq_tab, nza, naz) float64 power
 * with az periodic over 2 pi, za in [0, za_max] inclusive; nfreq_tab is 1 or nfreq.
 * order = beam_spline_opts["order"] (cpu/beams.py:69-74 -> pyuvdata az_za_map_coordinates ->
 * scipy.ndimage.map_coordinates): 0 = nearest node; 1 = bilinear; 2 .. 5 = interpolating B-spline of that
 * degree (the table is turned into spline coefficients on the device at upload; periodic in az,
 * mirrored in za).  1 and 3 have unrolled kernels, the others share one general path.
 * One order per handle.                                                                       */
int fv_sim_set_nbeams(fv_sim *h, int nbeams);
int fv_sim_set_beam_airy(fv_sim *h, int beam, double diameter);
/* The same dish with a complex factor per Jones slot, A[ax][feed] = jones_scale[ax][feed] 2 J1(x)/x
 * (jones_scale: 4 complex128 = 8 float64, row-major [ax][feed]; NULL = all ones), and a real factor on
 * the power beam, power_scale (2 J1(x)/x)^2.  How a third-party analytic Airy object is put on the device
 * in closed form: the host probes the object's own compute_response (cpu/beams.py:69-81 calls it per
 * slice), fits these factors and uses this entry only when every probe agrees to 1e-12 (e.g. pyuvdata's
 * AiryBeam: 1/sqrt(2) in every slot); otherwise the object is sampled onto a table.               */
int fv_sim_set_beam_airy_scaled(fv_sim *h, int beam, double diameter, const double *jones_scale,
                                double power_scale);
int fv_sim_set_beam_table(fv_sim *h, int beam, int nfreq_tab, int nza, int naz, double za_max,
                          const void *table, int order);

/* Beam pairs (prepare_beam_evaluation, cpu/beams.py:91-127): for pair p, beams (bi[p], bj[p]),
 * baseline indices idx[off[p] .. off[p+1]) and their `flipped` flags.  npairs = 1, bi=bj=0,
 * idx = 0..nbls-1, no flips is the beam_idx=None case.                                        */
int fv_sim_set_beam_pairs(fv_sim *h, int npairs, const int *bi, const int *bj, const int64_t *off,
                          const int *idx, const signed char *flipped);

/* Eigenbeam ("basis") mode, _compute_basis_visibilities (cpu_simulate.py:303-470): the handle's
 * beams 0..nbasis-1 are the K basis beams (E-field; polarized engine only, wrapper.py:280-283);
 * coefs is beam_coefs (nant, nbasis, nfreq) complex of the handle's precision; ant1/ant2 (nbls)
 * give each baseline's antenna indices into coefs (:920-921).  Replaces any beam pairs: every
 * (k <= l) term runs over all baselines without flips (:402-404) and is contracted as
 * conj(c[a1,k]) c[a2,l] V_kl + [l != k] conj(c[a1,l]) c[a2,k] V_kl^T (:461-468).
 * Call after fv_sim_set_array, fv_sim_set_freqs and the beams.                                */
int fv_sim_set_basis(fv_sim *h, int nant, int nbasis, int nfreq, const void *coefs, const int *ant1,
                     const int *ant2);

/* Two places where the reference's arithmetic differs from the exact symmetry of the visibilities (SURVEY
 * App. B Q1 / Q2).  on != 0 (the default): as the reference -- (1) a flipped baseline of a two-beam polarized pair
 * is evaluated at -b and conjugated, its 2 x 2 feed block NOT transposed (cpu_simulate.py:271,298); (2) the
 * eigenbeam (l, k) term reuses V_kl(b) transposed (:464-468), exact for real-valued basis beams only.
 * on == 0: (1) V_ji(b) = V_ij(-b)^H, conjugated and transposed; (2) V_lk(b) = conj(V_kl(-b))^T -- one more
 * gather at -b per off-diagonal term of complex basis beams.  Sticky on the handle; takes effect at the next run. */
int fv_sim_set_reference_compat(fv_sim *h, int on);

/* Source-axis chunking: the `for chunk in range(nchunks)` loop inside the reference's time loop
 * (cpu_simulate.py:939-946; visibilities accumulate with += over chunks, :1024,1069) and matvis'
 * source_buffer (cpu_simulate.py:693-704: the above-horizon arrays of a chunk hold
 * source_buffer x chunk size sources).  Every time step then processes the catalog in nchunks
 * consecutive pieces whose per-time device scratch (coordinates, bin sort, kernel weights, strengths)
 * is sized by one piece; the catalog itself stays resident.  A chunk with more sources above the
 * horizon than source_buffer allows fails the run (FV_ERR_ARG at the next synchronisation), as matvis
 * raises.  Defaults: nchunks = 1, source_buffer = 1.                                           */
int fv_sim_set_chunking(fv_sim *h, int nchunks, double source_buffer);

/* Run times [t0, t1) x freqs [f0, f1).  Result layout is the reference's FINAL layout
 * (cpu_simulate.py:850-854): polarized (nf_here, nt_here, 2, 2, nbls), else (nf_here, nt_here,
 * nbls), complex of the handle's precision.  out_on_device = 0: `out` is a host buffer (the
 * call synchronises); != 0: `out` is a device buffer and the call only enqueues work on the
 * handle's stream -- use fv_sim_sync().  A device `out` must be ordinary (coarse-grained) hipMalloc
 * memory on the handle's GPU: small 2-D grids are gathered with fp64 atomics compiled with
 * -munsafe-fp-atomics, which fine-grained or managed memory does not honour.
 * Bad input met on the device (source vectors that are NaN or not unit length, so that they fall
 * outside the planned grid; a type-1 entry overflow) fails the run at the next host
 * synchronisation -- this call for a host `out`, fv_sim_sync() otherwise -- with FV_ERR_ARG /
 * FV_ERR_INTERNAL; the output of that run is invalid.                                          */
int fv_sim_run(fv_sim *h, int t0, int t1, int f0, int f1, void *out, int out_on_device);
/* The same into a block INSIDE a larger host array in the final layout: channel f of the block starts
 * f * out_f_stride elements after `out` (0: contiguous, = fv_sim_run with a host `out`); its times are contiguous.
 * This is the reference's `vis[tc][..., fc] = future` (src/fftvis/cpu/cpu_simulate.py:843-847) without the copy: the
 * time blocks of a run that does not fit the device, and the ranks of a sharded run, deliver straight into their
 * slice of the result -- pinned in place run by run and filled from a copy stream while later time steps compute.
 * shared != 0: other processes write the rest of the array (one result in shared memory for all ranks of a node): the
 * pinning helper then only reads the block's pages when it touches them and registers nothing beyond its runs.        */
int fv_sim_run_into(fv_sim *h, int t0, int t1, int f0, int f1, void *out, int64_t out_f_stride, int shared);
int fv_sim_sync(fv_sim *h);

/* Introspection for bench/roofline: fills up to n doubles:
 * [0] spread launches, counted per (time, frequency group, beam pair) -- a gang launch that serves two
 * time steps counts twice, and a launch's transforms may run as several kernel launches --, [1] fine-grid cells written by spread (all trans, summed),
 * [2] source x trans visits, [3] cells moved through HBM by the pruned FFT passes,
 * [4] gathered footprints (targets x transforms, x 2 for packed transforms: read at s and -s), [5] above-horizon sources summed over times, [6] last n2x,
 * [7] last n2y, [8] last (na_x * 65536 + na_y), [9] kernel width w, [10] upsampling factor the
 * last run used, [11] largest above-horizon source count of any time step since the reset, [12] real flops of the FFT
 * passes priced as plain transforms (5 n2 log2 n2 per line transformed), [13] last n2 of the third dimension (1 for
 * 2-D runs), [14] last na of the third dimension, [15] height terms of the last run (K > 0: a non-coplanar array ran
 * as K 2-D transforms per slice, the expansion of exp(i z s_z) about the middle of the sources' height range; 0: no
 * expansion -- coplanar, or the 3-D transform), [16] lanes of the last type-3 run (2: consecutive time steps alternate
 * between two sets of scratch and grid buffers), [17] how they ran: 0 freely on two streams of equal priority (large
 * grids: the kernels of two time steps share the GPU, and kernel durations are those of kernels sharing it), 1
 * pipelined (big kernels in order on one stream, the next step's preparation beside them), 2 pipelined gangs,
 * [18] first LIGHT height term of the last run (terms k >= this ran on a second plan at a looser tolerance and
 * upsampling factor 1.25: they enter with weights 2 |J_k|; 0: none), [19] first term of a second, looser light class
 * (0: one class).                                                                                                   */
int fv_sim_stats(fv_sim *h, double *vals, int n);
int fv_sim_reset_stats(fv_sim *h);
/* HIP-event timing on the handle's stream (ms, summed since reset): [0] spread, [1] fft,
 * [2] interp, [3] strengths (beam + coherency), [4] rotate/sort, [5] number of spread launches
 * behind [0].  level 0: off; 1: spread only, events attached to the dispatches themselves (no extra
 * queue packets) for the spread launches of one time step in 16 of a run (the 9th: steady state) -- sampled because even
 * attached events idle the queue for a few us around a launch; cheap enough for a timed region; 2: every launch of every
 * family, bracketed by event records (adds ~10 us bubbles each; runs on a single stream); 3: as 1 but on
 * every spread launch (large grids, where a launch is hundreds of us and the bubble does not matter). */
int fv_sim_enable_timing(fv_sim *h, int level);
int fv_sim_timing(fv_sim *h, double *ms, int n);

/* ---- catalog exchange over RCCL (optional; SURVEY 8 b "fv_comm_init / fv_bcast_catalog", 8 e) ---------------------
 * For hosts that do not bring torch.distributed: the path's only exchange step is one broadcast of the catalog from the
 * rank that read it (the reference ships it to its Ray workers as a whole, cpu_simulate.py:711-847 via core/utils.py:
 * 122-187) -- positions to everyone, flux either whole (fv_bcast_catalog) or only the frequency columns of each rank's
 * block (fv_scatter_flux_columns: one packed ncclSend per rank in one group; C3 on 8 ranks: 32 instead of 250 MB per
 * rank).  Buffers are DEVICE pointers on the communicator's GPU; what arrives goes to fv_sim_set_sources(...,
 * on_device = 1).  Visibilities are never exchanged: every rank copies its own block out (fv_sim_run / fv_sim_run_into).
 * librccl.so.1 is opened on first use: without it these five entries fail with FV_ERR_INTERNAL and nothing else changes.
 *   fv_comm_unique_id : rank 0 fills FV_COMM_ID_BYTES bytes (ncclGetUniqueId); the host carries them to the other ranks
 *   fv_comm_init      : collective over the nranks processes (ncclCommInitRank), one process per GPU
 *   fv_bcast_catalog  : eq (3, nsrc) and flux as fv_sim_set_sources lays them out, as bytes; in place on every rank
 *   fv_scatter_flux_columns : flux_root_dev (nsrc, nfreq) entries of elem_bytes (8: float64, 4: float32, 64 / 32:
 *       2 x 2 complex) on the root; ranges = nranks pairs [f0, f1); out_dev (nsrc, f1 - f0) of THIS rank
 * Both transfers return after the data has arrived (the communicator's stream is synchronised).
 * Verified on hardware with one rank only (a one-GPU box cannot host two RCCL ranks); the Python host's
 * torch.distributed path (parallel.broadcast_catalog_device) is the one the multi-process tests cover.            */
#define FV_COMM_ID_BYTES 128
typedef struct fv_comm fv_comm;
int fv_comm_unique_id(void *id_bytes);
int fv_comm_init(fv_comm **c, int device, int rank, int nranks, const void *id_bytes);
int fv_comm_destroy(fv_comm *c);
int fv_bcast_catalog(fv_comm *c, int root, void *eq_dev, int64_t eq_bytes, void *flux_dev, int64_t flux_bytes);
int fv_scatter_flux_columns(fv_comm *c, int root, int64_t nsrc, int nfreq, int elem_bytes, const void *flux_root_dev,
                            const int *ranges, void *out_dev);

#ifdef __cplusplus
}
#endif
#endif /* FFTVIS_HIP_H */
