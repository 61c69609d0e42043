/*
 * kimi_hip.h -- C ABI of libkimi_hip.so, the MI355X (gfx950) implementation of the
 * kimimaro TEASAR hot path.
 *
 * Conventions
 *   - every entry point returns an int status (KH_OK == 0); kh_last_error() gives text.
 *     Nothing throws across the boundary (the reference's C++ `throw` at
 *     ext/skeletontricks/dijkstra_invalidation.hpp:54-58 would abort the process).
 *   - all volume pointers are DEVICE pointers (HBM resident), Fortran ordered:
 *     linear index = x + sx*(y + sy*z)  (ext/skeletontricks/skeletontricks.pyx:398).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous unless stated; the caller synchronises.
 *   - caller allocates everything; the library never frees caller memory.
 *   - labels are "connected component ids" 1..N (0 = background) as produced by
 *     kimimaro/utility.py:58-83; label_bytes is 2 or 4.
 *
 * The whole-volume design: because connected components are disjoint, the per-label
 * fields of the reference (DAF, PDRF, parents/dist, the invalidation mask -- all
 * per-crop numpy arrays in kimimaro/trace.py) live in ONE set of whole-volume arrays
 * shared by all labels, and the per-label kernels run one workgroup per label.
 */
#ifndef KIMI_HIP_H
#define KIMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_OK 0
#define KH_EINVAL 1      /* bad argument */
#define KH_EHIP 2        /* a HIP runtime call failed */
#define KH_ENODEVICE 3   /* no gfx950 device visible */

/* per-label status bits written by the device kernels (kh_label_t.status) */
#define KH_ST_OK 0u
#define KH_ST_QUEUE_OVERFLOW 1u   /* search work list overflow (scratch too small) */
#define KH_ST_HEAP_OVERFLOW 2u    /* invalidation heap overflow */
#define KH_ST_PATH_OVERFLOW 4u    /* path output buffer overflow */
#define KH_ST_NO_RAIL 8u          /* railroad: no rail reachable from a target */
#define KH_ST_PLATEAU 16u         /* a float-absorption plateau search found no exit (scratch exhausted) */
#define KH_ST_BAD_TARGET 32u      /* a target / root outside the label */

int kh_version(void);
/* copies the last error message of the calling thread, returns its length */
int kh_last_error(char* buf, int len);
/* number of visible devices whose arch is gfx950; 0 => every compute entry point fails loudly */
int kh_device_count(void);

/* ---- a1: edt.edt(labels, anisotropy, black_border) ---------------------------------
 * replaces: edt.edt as called at kimimaro/intake.py:178-183 and kimimaro/trace.py:112-117.
 * labels: u8/u16/u32 [sx,sy,sz]; out: f32 same shape; workspace: f32 of the same shape (ping-pong buffer, sx*sy*sz floats).
 * Squared distances are accumulated exactly as documented in oracle/kimi_oracle.c (ko_edt). */
int kh_edt(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz,
           float wx, float wy, float wz, int black_border,
           float* workspace, float* out, void* stream);

/* kh_edt for an array of `ndim` (1..3) dimensions, the trailing axes having extent 1: edt.edt on a 2-D array is a
 * 2-D transform -- with black_border the missing axes have no border (kimimaro/intake.py:568 calls it on the six
 * faces of the volume).  kh_edt == kh_edt_nd(ndim = 3).                                                       */
int kh_edt_nd(const void* labels, int label_bytes, int ndim, int64_t sx, int64_t sy, int64_t sz,
              float wx, float wy, float wz, int black_border,
              float* workspace, float* out, void* stream);

/* kh_edt with each pass bracketed by HIP events on `stream`; ms3 (HOST pointer) receives the
 * x / y / z pass durations in milliseconds.  Synchronises the stream (measurement only).       */
int kh_edt_timed(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz,
                 float wx, float wy, float wz, int black_border,
                 float* workspace, float* out, void* stream, float* ms3);

/* ---- preamble statistics on the device (fastremap.unique counts intake.py:198,
 * np.max(DBF) trace.py:100, first_label skeletontricks.pyx:307-326, x extent of
 * scipy.ndimage.find_objects utility.py:85-102) ------------------------------------
 * For every label id in [0, nlabels]: voxel count, max DBF, smallest linear index,
 * min/max x, and (yz_extent, nullable, 4 entries per label) [ymin, ymax, zmin, zmax].  All outputs
 * are device arrays of nlabels+1 entries, zero/identity initialised by the call.           */
int kh_label_stats(const void* labels, int label_bytes, const float* dbf, int64_t nvox, int64_t sx, int64_t sy,
                   int64_t nlabels, uint32_t* counts, float* dbf_max, uint32_t* first_index,
                   uint32_t* xmin, uint32_t* xmax, uint32_t* yz_extent, void* stream);

/* scatter the voxel indices of the selected labels into per-label lists.
 * slot_of_label: device int32[nlabels+1], -1 = label not selected, else its slot;
 * offsets: device u32[nslots] start of each slot's list in `lists`;
 * cursors: device u32[nslots] scratch (zeroed by the call).  Order inside a list is
 * unspecified (every consumer is order independent).                                  */
int kh_scatter_lists(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label,
                     int64_t nslots, const uint32_t* offsets, uint32_t* cursors, uint32_t* lists, void* stream);

/* 26-bit same-label connectivity mask per voxel (bit i = neighbour i of the order in
 * dijkstra_invalidation.hpp:60-124 is inside the volume and has the same non-zero
 * label).  The optional voxel_graph of the reference is ANDed in by the caller (NULL
 * in every config).                                                                   */
int kh_neighbor_mask(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz,
                     uint32_t* nbrmask, void* stream);

/* ---- per-label task descriptor (device array, one per selected label) -------------- */
typedef struct kh_label_t {
  uint32_t segid;        /* connected component id */
  uint32_t list_offset;  /* start of its voxel list */
  uint32_t count;        /* Nf */
  uint32_t xmin, xmax;   /* x extent of its bounding box (heap-order quirk, see paths.hip) */
  uint32_t source;       /* in: search source (linear index) for kh_edf_batch */
  uint32_t max_loc;      /* out: location of the farthest voxel */
  float max_val;         /* out: its distance */
  float M;               /* f32(1/dbf_max**1.01), kimimaro/trace.py:336 (host numpy) */
  uint32_t root;         /* root voxel (linear index) */
  uint32_t q_offset;     /* start of its slice of the work-list scratch (entries) */
  uint32_t q_capacity;   /* entries per work list */
  uint32_t heap_offset;  /* start of its invalidation heap slice (nodes) */
  uint32_t heap_capacity;
  uint32_t path_offset;  /* start of its slice of the path vertex buffer */
  uint32_t path_capacity;
  uint32_t tgt_offset;   /* manual targets: [tgt_offset, +n_before) before, then n_after after */
  uint32_t n_before, n_after;
  uint32_t max_paths;    /* 0 = unlimited (trace.py:214-215) */
  uint32_t n_paths;      /* out */
  uint32_t n_vertices;   /* out: vertices written to the path buffer */
  uint32_t status;       /* out: KH_ST_* bits */
  uint32_t stat_settled; /* out: sum of voxels touched by the railroad searches */
  uint32_t stat_heap_pushes; /* out (low 32 bits) */
  uint32_t cyc_target;   /* out: shader kilo-cycles spent in the target finder */
  uint32_t cyc_rail;     /* out: ... in railroad search + back-track */
  uint32_t cyc_inval;    /* out: ... in the invalidation flood, of which: */
  uint32_t cyc_pop;      /* out:   heap pops */
  uint32_t cyc_push;     /* out:   heap pushes */
  uint32_t cyc_fire;     /* out:   neighbour evaluation of live pops */
  /* soma mode (kimimaro/trace.py:119-134,160-168,246-251); all zero for ordinary labels */
  uint32_t soma_mode;    /* 1: root is the soma centre, one-off invalidation around it, paths are trimmed */
  float fsr;             /* free_space_radius of the DAF search = DBF[root] (trace.py:134) */
  float soma_radius;     /* dbf_max * soma_invalidation_scale + soma_invalidation_const (float32, trace.py:127) */
  float soma_scale;      /* soma_invalidation_scale */
  float soma_const;      /* soma_invalidation_const */
  /* order-free invalidation sweep (csrc/sweep.h); nlev == 0 switches it off for the label */
  uint32_t nlev;         /* in: number of distinct keys below the label's largest ball radius (levels it can touch) */
  float sweep_rmax;      /* in: a call whose largest radius exceeds this uses the heap emulation */
  uint32_t ev_offset;    /* in: start of its event arena, in units of 256 bytes */
  uint32_t ev_chunks;    /* in: event chunks in its arena */
  uint32_t ev_shift;     /* in: log2(slots per chunk), 4..7 (8 bytes per slot, slot 0 links the level's chunks) */
  uint32_t stat_sweep_calls;  /* out: invalidation calls tried by the sweep */
  uint32_t stat_sweep_bails;  /* out: ... of which fell back to the heap emulation */
  uint32_t stat_sweep_levels; /* out: levels processed */
  uint32_t stat_sweep_events; /* out: events processed (low 32 bits) */
  uint32_t stat_sweep_why;    /* out: OR of the bail reasons (sweep.h SW_BAIL_*) */
  uint32_t lev_window;   /* in: 0, or a power of two (>= 64) larger than the number of levels any event of this label can lie
                            ahead of the level being processed (= distinct keys within one 26-neighbour step of any key below
                            the label's radius, from the key table): the label then keeps lev_window level words, used round
                            robin, instead of nlev.  A bound that turns out too small costs speed, not results (the event
                            abandons the call to the heap emulation). */
  uint32_t ev_spill;     /* in: entries of the sweep's candidate-spill table at the front of the arena (0 or a power of two; 12 bytes
                            each, rounded up to 256 bytes): the fifth to eighth possible owner of a voxel (csrc/sweep.h) */
  uint32_t stat_ghost_calls;  /* out: calls of the sweep that left voxels undecided and went on with them as ghosts */
  uint32_t stat_rollbacks;    /* out: times the label rolled back to such a call and redid it by the heap emulation */
  /* KH_TRACE_FUSED_EDF: compute_pdrf's parameters (kimimaro/trace.py:315-356, the repeated-squaring branch) */
  uint32_t pdrf_log2e;   /* in: log2 of pdrf_exponent */
  float pdrf_scale;      /* in: pdrf_scale */
} kh_label_t;

/* ---- a4: dijkstra3d.euclidean_distance_field for a batch of labels ------------------
 * replaces the calls at kimimaro/trace.py:139-145 and :302-307.  For every task: the
 * distance field from task.source over its label (26-connected, anisotropic edge
 * lengths, f32 accumulation) is written into `field` at the label's voxels (other
 * voxels untouched); task.max_loc / max_val receive the farthest voxel (ties ->
 * smallest linear index).  queues: device u32[...] scratch addressed by q_offset (4 lists
 * of q_capacity >= count entries per task); qstate: u8 per voxel, all zero on entry and on exit
 * (two membership bits per voxel keep every work list bounded by the label size).
 * mode 0: source = task.source.
 * mode 1 (find_root, trace.py:291-308): only tasks with root == 0xFFFFFFFF run; afterwards
 *         task.root = max_loc.   mode 2 (DAF, trace.py:139-145): source = task.root.
 * Bits 8.. of `mode`: threads per label (a multiple of 64 up to 1024; 0 = 512).                  */
int kh_edf_batch(kh_label_t* tasks, int ntasks, int mode, const uint32_t* lists, const uint32_t* nbrmask,
                 int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                 float* field, uint8_t* qstate, uint32_t* queues, void* stream);

/* ---- voxel_graph of dijkstra3d.* / roll_invalidation_ball_inside_component (kimimaro/trace.py:139-145,155,167,240-242,257;
 * dijkstra_invalidation.hpp:126-191): nbrmask[v] &= the directions the caller's connectivity word graph[v] allows (cc3d's bit layout,
 * read at the CURRENT voxel like the reference does).  Every search and the invalidation take their neighbourhoods from nbrmask, so
 * after this call they honour the graph.  corner_gate (nullable, u8 per voxel): bit j = the yz diagonal that corner entry 18 + j
 * degenerates into at an x face of the array exists AND the corner's own bit is set (dijkstra_invalidation.hpp:116-123 + :182-190);
 * kh_invalidate_ball takes it so that the heap emulation pushes the same duplicates as the reference there.                    */
int kh_apply_voxel_graph(uint32_t* nbrmask, const uint32_t* graph, int64_t nvox, uint8_t* corner_gate, void* stream);

/* ---- a3+a5: zero2inf / inf2zero / compute_pdrf fused, whole volume ------------------
 * replaces kimimaro/trace.py:138,146,148 (skeletontricks.pyx:177-224, trace.py:315-356).
 * For every voxel of a selected label: daf *= 1/max_daf (max_daf = task.max_val, skipped
 * when 0), pdrf = ((1 - dbf*M)^(2^log2_exponent)) * scale + daf, each op rounded to f32.
 * Voxels of unselected labels / background get pdrf = +inf.                           */
/* Exponents that are not a power of two take the reference's np.power branch (kimimaro/trace.py:346-347), whose rounding
 * is that of the host's numpy (libm / SVML powf -- machine dependent, so no device powf can promise the same bits).
 * The call is then made twice around the host's own np.power on the pdrf buffer:
 *   log2_exponent = KH_PDRF_BASE    pdrf = 1 - dbf*M for the selected voxels (+inf elsewhere), daf untouched;
 *   (host: np.power(pdrf, exponent, out=pdrf) in float32)
 *   log2_exponent = KH_PDRF_FINISH  pdrf = pdrf*scale + daf/max_daf, daf normalised in place -- the tail of the normal call. */
#define KH_PDRF_BASE (-1)
#define KH_PDRF_FINISH (-2)
#define KH_PDRF_KEEP_OTHERS 0x100   /* ORed into a log2_exponent >= 0: the voxels of unselected labels are left as they are (instead of
                                       +inf) -- the selected labels are only a part of what another launch computes into the same volume */
int kh_pdrf(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label,
            const kh_label_t* tasks, const float* dbf, float* daf, int log2_exponent, float scale,
            float* pdrf, void* stream);

/* ---- a6..a11: the per-label TEASAR path loop ---------------------------------------
 * replaces kimimaro/trace.py:196-267 (compute_paths) with its callees
 * CachedTargetFinder.find_target (skeletontricks.pyx:1008-1045), dijkstra3d.railroad
 * (trace.py:240-242), roll_invalidation_ball_inside_component (skeletontricks.pyx:373-418
 * -> dijkstra_invalidation.hpp:239-332) and the rail edits trace.py:220,261-263.
 * alive: u8 per voxel, 1 for the voxels of selected labels (mutated: invalidation);
 * pdrf: mutated (rails are zeroed); dist: f32 scratch volume, must be +inf on entry and
 * is +inf again on exit; list_daf: DAF gathered in list order (target finder keys).
 * Paths are written as linear indices, rail end first, into path_vertices with
 * path_lengths (one u32 per path, same slice offsets /capacity as the vertices).
 * qstate as for kh_edf_batch.  heap_nodes: scratch for the invalidation heaps, 16 bytes per node
 * (16-byte aligned), each label owning nodes [heap_offset, heap_offset + heap_capacity).
 * Invalidation runs as the order-free level sweep of csrc/sweep.h whenever that certifies the call (the result is
 * then independent of the heap's tie order, hence equal to the reference's), and as the exact emulation of
 * std::priority_queue otherwise.  The sweep needs: level_rank = u32 [ra, rb, rc] table (x fastest) of the rank of
 * the key of offset (a, b, c) among the distinct keys (kh_level_keys + sort/unique by the caller), cstate = one
 * zeroed u64 per voxel (zero again on exit), event_arena = 256-byte aligned scratch addressed by ev_offset /
 * ev_chunks / ev_shift / nlev / lev_window of each task; max_nlev = the number of level words every workgroup of the launch
 * gets in LDS (<= KH_SWEEP_LDS_LEVELS): a task keeps its words there when its lev_window (if non-zero) or else its nlev fits;
 * a task for which neither does runs all its invalidations as the heap emulation.  A task's arena = the candidate-spill table,
 * a free stack of ev_chunks u32 (each rounded up to 256 bytes), then the chunks.  level_rank == NULL switches the sweep off.
 * sched = one u32 per voxel, 0xFFFFFFFF on entry: the sweep's filter words (csrc/sweep.h) -- the pending deadline of a live voxel
 * (with it a voxel is handed ~1.3 events per call instead of ~13), 0 once the voxel is dead (the sweep reads its neighbours' state
 * from these words, nine rows of three per event: the word BEFORE sched[0] and the word behind the last one must be readable).
 * NULL switches the sweep off.
 * Integer levels (round 6): level_rank == NULL and ra > 0 selects them -- (ra, rb, rc) = (wx^2, wy^2, wz^2) / gcd for an integral
 * anisotropy whose squared distances stay exact in float over the balls' reach (kimimaro_amd.engine.int_key_mode checks it): the
 * level of a key is then the integer key^2 / gcd itself, no table, no float operation per neighbour; task.nlev / lev_window count
 * such integers.  level_rank == NULL and ra == 0: no sweep.
 * Ghosts: journal (nullable) = u32 scratch, 2 * q_capacity entries per task at 2 * q_offset; rail_save (nullable)
 * = f32 scratch laid out like path_vertices.  With both, a call of the sweep that leaves voxels undecided goes on with them as
 * "ghosts" instead of running the heap emulation at once; the label rolls back to that call and redoes it exactly only if a
 * ghost would influence the loop.  Results are identical either way (KH_TRACE_NO_GHOSTS switches it off for A/B runs,
 * KH_TRACE_GHOST_PARANOID rolls every such call back at once -- a test of the roll-back itself).  Each task's arena starts with
 * its candidate-spill table (ev_spill entries of 12 bytes, rounded up to 256 bytes), then the free stack, then the chunks.
 * flags: KH_TRACE_PROFILE also fills cyc_pop / cyc_push / cyc_fire (slower kernel variant); KH_TRACE_HEAP_PRIO see below.
 * fix_branching = 0 selects the parental-field variant (trace.py:155,244): one weighted Dijkstra
 * from the root, paths returned root -> target.                                                  */
#define KH_TRACE_PROFILE 1
#define KH_TRACE_HEAP_PRIO 2   /* the wave running the heap emulation raises its issue priority (several volumes in flight) */
#define KH_TRACE_THREADS_64 4  /* workgroups of 64 threads (one wave per label, up to 12 labels per CU) instead of 256 */
#define KH_TRACE_THREADS_128 8 /* workgroups of 128 threads */
#define KH_TRACE_NO_GHOSTS 16  /* undecided voxels abandon the call to the heap emulation at once (rounds 2-4) */
#define KH_TRACE_GHOST_PARANOID 32  /* roll back after every call that made a ghost (tests) */
#define KH_TRACE_VOXEL_GRAPH 128    /* nbrmask went through kh_apply_voxel_graph (voxel_graph= of kimimaro.trace.trace): the edges are
                                       one-way, the predecessor walks read the step u -> v from u's word; corner_gate = the array that
                                       call filled (NULL without a graph) */
#define KH_TRACE_BIG_LDS_HEAP 64    /* two chunks (8191 nodes, 128 KiB) of every label's invalidation heap live in LDS: one workgroup
                                       per CU, for a launch of the few largest labels whose heap emulation sets the wall clock */
#define KH_TRACE_SCRATCH_POOL 256   /* heap_nodes is ONE pool for the launch instead of a slice per label: node 0 = {u32 nodes handed out so
                                       far (the caller sets 1), u32 nodes in the pool, -, -}; a label takes its heap (heap_capacity nodes) when
                                       it first runs the heap emulation and its ghost journal ((2 * q_capacity + 3) / 4 nodes) when it first
                                       makes a ghost; `journal` is ignored, heap_offset too.  A label the pool cannot serve ends with
                                       KH_ST_HEAP_OVERFLOW (trace it again with a slice of its own) or goes on without ghosts. */
#define KH_TRACE_FUSED_EDF 512      /* the label's workgroup runs find_root, the DAF search and compute_pdrf itself before its path loop
                                       (what kh_edf_batch modes 1 and 2, kh_gather_f32 and kh_pdrf do for all labels at once): a task with
                                       root == 0xFFFFFFFF gets its root from the search from task.source; list_daf and pdrf are OUTPUTS for
                                       the tasks' voxels (pdrf elsewhere is not read), dist serves as the searches' field; task.M,
                                       pdrf_log2e, pdrf_scale, fsr as for kh_pdrf / kh_edf_batch.  Only the power-of-two exponents. */
#define KH_SWEEP_LDS_LEVELS 16384   /* level words kept in LDS up to this many levels per label */
int kh_trace_paths(kh_label_t* tasks, int ntasks, const uint32_t* lists, float* list_daf,
                   const uint32_t* nbrmask,
                   int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                   const float* dbf, float* pdrf, float* dist, uint8_t* alive, uint8_t* qstate,
                   const uint32_t* manual_targets, float scale, float constant,
                   uint32_t* queues, void* heap_nodes,
                   uint32_t* path_vertices, uint32_t* path_lengths,
                   const uint32_t* level_rank, int64_t ra, int64_t rb, int64_t rc, int64_t max_nlev,
                   uint64_t* cstate, uint32_t* sched, void* event_arena, uint32_t* journal, float* rail_save,
                   const uint8_t* corner_gate, int flags, int fix_branching, void* stream);

/* keys[a + ra*(b + rb*c)] = the flood's key of the voxel offset (a, b, c): sqrt(fl(fl((wx*a)^2 + (wy*b)^2) + (wz*c)^2)),
 * float operation order of dijkstra_invalidation.hpp:310-316.  Device array of ra*rb*rc floats.              */
int kh_level_keys(int64_t ra, int64_t rb, int64_t rc, float wx, float wy, float wz, float* keys, void* stream);

/* ---- a9 on its own: kimimaro.skeletontricks.roll_invalidation_ball_inside_component(labels, DBF, scale, const,
 * anisotropy, path) (skeletontricks.pyx:373-418 -> dijkstra_invalidation.hpp:239-332) for ONE object, the device routine
 * of the path loop behind its own entry point.  task: a kh_label_t of the object (count, xmin, xmax, list_offset,
 * q_offset / q_capacity, heap_offset / heap_capacity and the sweep fields as for kh_trace_paths); alive: u8 per voxel,
 * 1 for the object's voxels that are still valid (mutated in place like the reference's `labels`); path: u32 linear
 * indices of the path vertices; *invalidated (device int64) = number of voxels invalidated by the call.
 * The ball radius of vertex v is f32(f32(scale * dbf[v]) + constant).  Sweep arguments as for kh_trace_paths.
 * corner_gate: NULL, or the array kh_apply_voxel_graph filled (voxel_connectivity_graph=, skeletontricks.pyx:380,405-416).      */
int kh_invalidate_ball(kh_label_t* task, const uint32_t* lists, const uint32_t* nbrmask,
                       int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                       const float* dbf, uint8_t* alive, uint32_t* queues, void* heap_nodes,
                       const uint32_t* path, int64_t npath, float scale, float constant,
                       const uint32_t* level_rank, int64_t ra, int64_t rb, int64_t rc, int64_t max_nlev,
                       uint64_t* cstate, uint32_t* sched, void* event_arena, const uint8_t* corner_gate, int64_t* invalidated,
                       void* stream);

/* ---- a7 / a8 on their own: one search on one object (the path loop has them inside kh_trace_paths).
 * field: the weight volume (PDRF; +inf outside the object); dist: f32 volume, +inf on entry; qstate / queues as for
 * kh_edf_batch; task: the object's kh_label_t (q_offset / q_capacity, count).
 *   mode 0  dijkstra3d.railroad(field, source) (kimimaro/trace.py:240-242): path from `source` to the nearest voxel of
 *           weight 0, written rail end first; dist is +inf again on exit.
 *   mode 1  the search of dijkstra3d.parental_field(field, source) (trace.py:155): leaves the distances in `dist`.
 *   mode 2  dijkstra3d.path_from_parents(parents, target) (trace.py:244) on that `dist`: path source -> target.
 * *path_length (device u32) = vertices written (0 on failure, see task.status).  Ties between equally short paths are
 * resolved by the canonical predecessor rule of DESIGN.md 3.3 (dijkstra3d's own tie order is not observable: its
 * source is absent from the reference tree).  voxel_graph != 0: nbrmask went through kh_apply_voxel_graph (voxel_graph= of the
 * dijkstra3d calls): one-way edges, the predecessor walk reads the step u -> v from u's word.                          */
int kh_path_search(kh_label_t* task, int mode, const uint32_t* lists, const uint32_t* nbrmask,
                   int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                   const float* field, float* dist, uint8_t* qstate, uint32_t* queues,
                   uint64_t source, uint64_t target, uint32_t* path, int64_t path_capacity, uint32_t* path_length,
                   int voxel_graph, void* stream);

/* ---- a7 as ARRAYS: dijkstra3d.parental_field(field, source) -> parents and dijkstra3d.path_from_parents(parents, target)
 * (kimimaro/trace.py:155, 244).  The reference's caller edits the parents array between the two calls
 * (`parents[tuple(root)] = 0`, trace.py:220), so it has to be a plain array the caller owns.
 * kh_parental_field: the search of kh_path_search mode 1 (distances left in `dist`; task / lists / nbrmask / qstate / queues as
 *   there) followed by the canonical predecessor rule on every voxel of the object (oracle ko_parental_field):
 *   parents[v] = linear index of pred(v) + 1, 0 = none (the source, unreachable voxels, everything outside the object).
 *   parents: u32 [sx*sy*sz], zeroed by the call.  A voxel whose achieving neighbours all lie at its own distance (float
 *   absorption plateau) cannot be expressed as a pointer: task.status gets KH_ST_PLATEAU (like the oracle's KO_EPLATEAU).
 * kh_path_from_parents: the pointer chase target -> ... -> a voxel whose entry is 0, written source first; *path_length = 0
 *   when path_capacity is too small (or the caller's edits made a cycle).                                               */
int kh_parental_field(kh_label_t* task, const uint32_t* lists, const uint32_t* nbrmask, int64_t sx, int64_t sy, int64_t sz,
                      const float* field, float* dist, uint8_t* qstate, uint32_t* queues, uint64_t source,
                      uint32_t* parents, int voxel_graph, void* stream);
int kh_path_from_parents(const uint32_t* parents, int64_t nvox, uint64_t target, uint32_t* path, int64_t path_capacity,
                         uint32_t* path_length, void* stream);

/* ---- a3 / a5 / a6 as stand-alone operations (the path loop has them fused: kh_pdrf, kh_trace_paths) ----------------
 * kh_zero2inf / kh_inf2zero: skeletontricks.zero2inf / inf2zero (skeletontricks.pyx:177-224), in place.
 * kh_pdrf_field: compute_pdrf (kimimaro/trace.py:315-356) on every element: out = ((1 - dbf*M)^(2^log2_exponent)) * scale
 *   (+ daf / max_daf, with daf normalised in place, when max_daf != 0); every operation rounded to f32.  KH_PDRF_BASE /
 *   KH_PDRF_FINISH as for kh_pdrf (the two halves around the host's np.power for other exponents).
 * kh_target_max: CachedTargetFinder.find_target (skeletontricks.pyx:1008-1045) over a voxel list with its DAF values:
 *   *out (device u64) = 1<<63 | DAF bits << 32 | voxel of the valid (alive != 0) voxel with the largest DAF, ties ->
 *   largest index; 0 when no voxel is valid.                                                                       */
int kh_zero2inf(float* f, int64_t n, void* stream);
int kh_inf2zero(float* f, int64_t n, void* stream);
int kh_pdrf_field(const float* dbf, float* daf, int64_t n, float M, int log2_exponent, float scale, float max_daf,
                  float* out, void* stream);
int kh_target_max(const uint32_t* list, const float* list_daf, const uint8_t* alive, int64_t n, uint64_t* out, void* stream);
/* kh_find_target: the legacy skeletontricks.find_target(labels, PDRF) (skeletontricks.pyx:331-367): the first voxel in the
 *   reference's scan (x outermost, z innermost) whose field value is the maximum over the mask (strict >, from -inf).
 *   labels u8 / field f32 [sx,sy,sz] Fortran ordered; *out (device u64) = 0 when the mask is empty (the reference returns
 *   (-1,-1,-1)), else orderable value bits << 32 | (0xFFFFFFFF - (x*sy + y)*sz - z).
 * kh_first_label: skeletontricks.first_label (skeletontricks.pyx:307-326): *out (device u64) = smallest linear index of a
 *   non-zero voxel, ~0 when there is none (the reference returns None).                                              */
int kh_find_target(const uint8_t* labels, const float* field, int64_t sx, int64_t sy, int64_t sz, uint64_t* out, void* stream);
int kh_first_label(const uint8_t* labels, int64_t n, uint64_t* out, void* stream);

/* small helpers used by the host mirror */
int kh_fill_f32(float* p, int64_t n, float v, void* stream);
int kh_fill_u8(uint8_t* p, int64_t n, int v, void* stream);
int kh_gather_f32(const float* src, const uint32_t* idx, int64_t n, float* out, void* stream);
/* alive[v] = 1 where slot_of_label[label[v]] >= 0 else 0 */
int kh_init_alive(const void* labels, int label_bytes, int64_t nvox, const int32_t* slot_of_label,
                  uint8_t* alive, void* stream);

/* ---- a10: roll_invalidation_cube (skeletontricks.pyx:766-836 -> skeletontricks.hpp:42-155)
 * not on the reference's own skeletonize() call graph, exported and tested there
 * (automated_test.py:632-825).  mask: u8 device [sx,sy,sz], mutated; path: device u64
 * linear indices; *invalidated (device int64) receives the count.                      */
int kh_invalidate_cube(uint8_t* mask, const float* dbf, int64_t sx, int64_t sy, int64_t sz,
                       float wx, float wy, float wz, const uint64_t* path, int64_t npath,
                       float scale, float constant, int64_t* invalidated, void* stream);

/* ---- preamble row f1: 26-connected multi-label connected components on the DEVICE.
 * replaces cc3d.connected_components + fastremap.renumber/refit as called at kimimaro/utility.py:58-83.
 * labels: u8/u16/u32/u64 [sx,sy,sz]; out: u32 component ids 1..N by first appearance in the F-order raster
 * (the numbering of kh_host_ccl26); parent: u32 scratch [nvox]; chunk_counts: u32 scratch [ceil(nvox/1024)];
 * representative: u32 [nvox+1 worst case; N+1 used], representative[id] = smallest linear index of the
 * component (skeletontricks.get_mapping, skeletontricks.pyx:490-525, reads the original label there);
 * *ncomponents (device u32) = N.  out16 (nullable, u16 [nvox]): filled with the same ids when N < 65536
 * (fastremap.refit, kimimaro/utility.py:79), untouched otherwise.                                     */
int kh_ccl26(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz, uint32_t* parent,
             uint32_t* chunk_counts, uint32_t* out, uint32_t* representative, uint32_t* ncomponents,
             uint16_t* out16, void* stream);

/* ---- skeletonize(voxel_graph=) (kimimaro/intake.py:66,162,174-183; utility.py:73-75).  PARITY UNPINNED: cc3d and edt are
 * third-party packages whose sources are absent from the reference tree.
 * kh_ccl26_graph replaces cc3d.color_connectivity_graph(voxel_graph, connectivity=26) followed by `cc_labels *= all_labels > 0`:
 * two foreground voxels are joined iff they are 26-neighbours and the graph word (cc3d's bit layout, u32 per voxel) of the later
 * one in the raster allows the step to the earlier one.  Arguments and numbering as kh_ccl26.
 * kh_edt_graph_cells / kh_edt_graph_sample are the two ends of edt.edt(labels, voxel_graph=): cells (u8 [2sx, 2sy, 2sz], F order)
 * receives the doubled image -- voxel (x, y, z) at cell (2x, 2y, 2z); the cell towards +x / +y / +z is foreground iff the voxel is
 * and bit 0 / 2 / 4 of its graph word is set (without black_border: also behind the last voxel of an axis); the other cells of the
 * voxel's 2 x 2 x 2 block follow the voxel -- on which the caller runs kh_edt(label_bytes = 1) with HALF the voxel pitch;
 * kh_edt_graph_sample reads that transform back at the voxel cells.  8 * nvox < 2^32.                                          */
int kh_ccl26_graph(const void* labels, int label_bytes, const uint32_t* graph, int64_t sx, int64_t sy, int64_t sz,
                   uint32_t* parent, uint32_t* chunk_counts, uint32_t* out, uint32_t* representative, uint32_t* ncomponents,
                   uint16_t* out16, void* stream);
int kh_edt_graph_cells(const void* labels, int label_bytes, const uint32_t* graph, int64_t sx, int64_t sy, int64_t sz,
                       int black_border, uint8_t* cells, void* stream);
int kh_edt_graph_sample(const float* fine, int64_t sx, int64_t sy, int64_t sz, float* out, void* stream);

/* ---- f3: binary hole filling, replaces fill_voids.fill(img, in_place=True, return_fill_count=True) as
 * called at kimimaro/trace.py:109 (third-party, source absent): a background voxel (mask == 0) stays
 * background iff a 6-connected background path joins it to a face of the array.  mask/out: u8 [nvox]
 * (out may not alias mask); parent: u32 [nvox] scratch; open: u8 [nvox] scratch; *filled (device i64) =
 * number of voxels that changed.                                                                    */
int kh_fill_voids(const uint8_t* mask, int64_t sx, int64_t sy, int64_t sz, uint32_t* parent, uint8_t* open,
                  uint8_t* out, int64_t* filled, void* stream);
/* the same for an array of `ndim` (1..3) dimensions, the axes beyond it having extent 1: they are not axes and have no faces -- the
 * border of a 2-D image is its outline (fill_voids.fill on the six faces of a crop, kimimaro/intake.py:655-666, fix_avocados).
 * kh_fill_voids == kh_fill_voids_nd(ndim = 3).                                                                      */
int kh_fill_voids_nd(const uint8_t* mask, int ndim, int64_t sx, int64_t sy, int64_t sz, uint32_t* parent, uint8_t* open,
                     uint8_t* out, int64_t* filled, void* stream);

/* ---- kh_ccl26 on HOST memory (used for the 2-D faces of fix_borders and as a cross-check),
 * restating cc3d.connected_components as called at kimimaro/utility.py:74-77.
 * Returns the number of components (ids 1..N by first appearance in F-order raster).   */
int64_t kh_host_ccl26(const void* labels, int label_bytes, int64_t sx, int64_t sy, int64_t sz,
                      uint32_t* out);

/* ---- row f2 (host side): skeletontricks.find_border_targets (skeletontricks.pyx:591-647 with
 * compute_centroids :528-588 and compute_tiebreaker_maxima :650-760) on HOST memory.
 * dt (f32) and cc (u32) are [sx,sy] Fortran ordered planes; for each component id with a maximum
 * out_xy[2*id..2*id+1] receives (x, y) and `order` the ids in dict-insertion order.
 * Returns the number of ids, -1 on allocation failure.                                  */
int64_t kh_host_find_border_targets(const float* dt, const uint32_t* cc, int64_t sx, int64_t sy,
                                    float wx, float wy, int64_t nlab, float* out_xy, int32_t* order);

/* ---- row a12 (host side): Skeleton.from_path per path + Skeleton.simple_merge + consolidate (kimimaro/trace.py:182-184) for
 * every component ("slot") of a result group in one call, outside the interpreter.  voff / loff [nslots+1]: first path vertex /
 * first path of every slot in locs (linear voxel indices x + sx (y + sy z), the paths of a slot back to back) / lens (vertices
 * per path); radii: the DBF at every path vertex.  Output, slot by slot: out_verts [N,3] = the slot's unique vertices that an
 * edge refers to, as voxel coordinates in f32, sorted lexicographically by (x, y, z) like np.unique(axis=0); out_radii [N] = the
 * radius of each vertex's first occurrence; out_edges [M,2] = rows (lo, hi), lo != hi, unique, sorted, indices local to the slot;
 * vstart / estart [nslots+1] = where every slot's vertices / edges start.  The caller sizes out_verts / out_radii for sum(lens)
 * vertices and out_edges for sum(lens) edges.  Returns N, -1 on allocation failure.                                             */
int64_t kh_host_consolidate_paths(int64_t nslots, const int64_t* voff, const int64_t* loff, const uint32_t* locs,
                                  const uint32_t* lens, const float* radii, int64_t sx, int64_t sy, int64_t sz,
                                  float* out_verts, float* out_radii, uint32_t* out_edges, int64_t* vstart, int64_t* estart);

/* ---- row a12 (host side): the merge of a label's components into one skeleton, kimimaro/intake.py:587-593
 * (Skeleton.simple_merge(...).consolidate() per original label) for ALL labels of a volume in one call, outside the
 * interpreter.  The components of a label are disjoint voxel sets whose vertices are sorted lexicographically by
 * (x, y, z) already (kimimaro_amd.intake.consolidate_paths_batch), so consolidate() is a re-sort on integer keys.
 * Parts are given label by label, in component order: part_of_label[l] .. part_of_label[l+1] are label l's parts;
 * vstart / estart [nparts+1]: first vertex / edge of every part in verts [N,3] (voxel coordinates as f32) / radii [N] /
 * edges [M,2] (indices local to the part).  Output, label by label at the same offsets: out_verts = sorted vertices
 * times (ax, ay, az) in f32 (intake.py:513), out_radii, out_edges [M,2] = rows (lo, hi) sorted lexicographically, indices
 * local to the LABEL.  Returns 0, -1 on allocation failure.                                                         */
int64_t kh_host_merge_components(int64_t nlabels, const int64_t* part_of_label, const int64_t* vstart, const int64_t* estart,
                                 const float* verts, const float* radii, const uint32_t* edges, int64_t sy, int64_t sz,
                                 float ax, float ay, float az, float* out_verts, float* out_radii, uint32_t* out_edges);

#ifdef __cplusplus
}
#endif
#endif
