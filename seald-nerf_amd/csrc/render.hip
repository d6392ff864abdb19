// Device-driven inference loop for one frame (dnerf/renderer.py:333-381 of the reference, `-O` numerics).
//
// The reference re-enters Python after every loop iteration: `rays_alive[rays_alive >= 0]` needs the survivor count on
// the host (a device->host sync), and `n_step` is derived from it.  Here n_alive, n_step, the step budget, the
// ping-pong side of the alive list and the per-iteration live-sample counter live in a 32-byte device record that
// the kernels of an iteration read and a one-thread kernel advances, so the host can enqueue iteration k+1 while
// iteration k runs; it only needs an UPPER BOUND of n_alive to size the grids (the survivor count it read back one
// iteration earlier), and learns that the loop is over one iteration late (that iteration is a no-op).
// The schedule -- n_step = clamp(N // n_alive, 1, 8), stop at max_steps -- and every sample are the reference's.
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdn_common.h"
#include "sdn_internal.h"

namespace sdn_int {
int render_begin(const SdnRenderCtx *c, void *mailbox, uint32_t frame_tag, hipStream_t st) {
    if (c->aabb) {  // nears / fars of this frame's rays, on the frame's stream (one Python round trip less per frame)
        int rc0 = sdn_near_far_from_aabb(c->rays_o, c->rays_d, c->aabb, c->N, c->min_near, (float *)c->nears, (float *)c->fars, st);
        if (rc0) return rc0;
    }
    int rc = loop_begin(c->N, c->max_steps, c->nears, c->alive_a, c->rays_t, c->weights_sum, c->depth, c->image, c->state, c->live_counts,
                        c->n_counters, mailbox, frame_tag, c->rays_tend, st);
    if (rc) return rc;
    if (c->H == 128 && c->C == 1) {
        const uint32_t nf = c->n_group_frames > 1 ? c->n_group_frames : 1u;
        bool kept = true;
        for (uint32_t f = 0; f < nf; f++) kept = kept && c->frame_cull[f] != nullptr;
        if (kept) rc = copy_cull(c->frame_cull, nf, (uint32_t *)c->cull_bits, st);
        else rc = c->n_group_frames > 1 ? build_cull_group(frame_sel(c), (uint32_t *)c->cull_bits, st) : build_cull(c->bitfield, (uint32_t *)c->cull_bits, st);
        // iteration 0 on the rays that pass the exact cull test (same samples, same trace; raymarching.hip k_cull_start)
        if (!rc && c->rays_tend)
            rc = loop_cull_start(c->N, c->rays_o, c->rays_d, c->nears, c->fars, c->bound, c->dt_gamma, c->C, c->H, (const uint32_t *)c->cull_bits, frame_sel(c),
                                 c->alive_a, c->alive_b, (float *)c->rays_tend, c->state, (uint32_t *)c->block_totals, c->trace + 2 * (size_t)c->n_counters + 8,
                                 c->trace, c->max_steps, c->sigmas, st);   // (sigmas: unused until the first field launch -- holds the per-ray jump targets)
    }
    return rc;
}
}  // namespace sdn_int

namespace {
std::atomic<int> g_timed_kernel{0};   // sdn_render_time_kernel: 0 = the field launch, 1 = the marcher launch of an iteration
}

extern "C" {

int sdn_render_time_kernel(int which) {
    if (which < 0 || which > 1) return SDN_E_BADARG;
    g_timed_kernel.store(which);
    return 0;
}

int sdn_render_finish(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream);
int sdn_render_step_f16_ev(const SdnRenderCtx *c, uint32_t bound_alive, void *ev_field_begin, void *ev_field_end, void *stream);

// SealD edit hooks of an iteration (no-ops without ctx->seal): samples back to their origin before the field network, colours of
// the mapped samples after it.  m_slots bounds the slots the marcher wrote this iteration.
static int seal_map(const SdnRenderCtx *c, uint32_t m_slots, hipStream_t st) {
    const SdnSealBox *s = c->seal;
    if (!s) return 0;
    if (!c->seal_mask) return SDN_E_BADARG;
    if (s->has_map_source) {
        if (!s->scratch) return SDN_E_BADARG;
        return sdn_seal_bbox_map_source(c->xyzs, c->dirs, m_slots, s->bounds, s->n_bounds, s->tris, s->n_tris, s->test_dir, s->tinv, s->rinv,
                                        s->scale, s->center, s->source_bound, s->map_source, (uint32_t *)((char *)s->scratch + 16), c->seal_mask,
                                        c->live_idx, (const uint32_t *)c->live_counts, c->state, st);
    }
    return sdn_seal_bbox_map(c->xyzs, c->dirs, m_slots, s->bounds, s->n_bounds, s->tris, s->n_tris, s->test_dir, s->tinv, s->rinv, s->scale,
                             s->center, c->seal_mask, st);
}
static int seal_color(const SdnRenderCtx *c, uint32_t m_slots, hipStream_t st) {
    const SdnSealBox *s = c->seal;
    if (!s) return 0;
    int rc = 0;
    if (s->modify_hsv) rc = sdn_seal_modify_hsv(c->rgbs, c->seal_mask, m_slots, s->hsv[0], s->hsv[1], s->hsv[2], st);    // map_color's order: hsv, then rgb
    if (!rc && s->modify_rgb) {
        if (!s->scratch) return SDN_E_BADARG;
        rc = sdn_seal_modify_rgb(c->rgbs, c->seal_mask, m_slots, s->rgb[0], s->rgb[1], s->rgb[2], s->rgb_light_offset, s->scratch, c->live_idx,
                                 (const uint32_t *)c->live_counts, c->state, st);
    }
    return rc;
}

// the fused field network on this iteration's samples (live list, count on the device): the `-O` kernel, or the fp32 one (ctx->field_f32)
static int launch_field(const SdnRenderCtx *c, uint32_t m_bound, uint32_t expect_points, hipStream_t st) {
    if (c->field_f32 == 2)
        return sdn_int::field_forward_f32x3(c->xyzs, c->dirs, c->live_idx, (const uint32_t *)c->live_counts, c->state, m_bound,
                                            (const float *)c->field_weights, c->field_bias0, (const float *)c->grid_table, c->grid_offsets, c->grid_S,
                                            c->grid_H, c->bound, c->density_scale, c->zero_deform, c->sigmas, c->rgbs, nullptr,
                                            c->n_group_frames > 1 ? c->slot_frame : nullptr, c->n_group_frames > 1 ? c->n_group_frames : 1u, st);
    if (c->field_f32)
        return sdn_int::field_forward_f32(c->xyzs, c->dirs, c->live_idx, (const uint32_t *)c->live_counts, c->state, m_bound,
                                          (const float *)c->field_weights, c->field_bias0, (const float *)c->grid_table, c->grid_offsets, c->grid_S,
                                          c->grid_H, c->bound, c->density_scale, c->zero_deform, c->sigmas, c->rgbs, nullptr,
                                          c->n_group_frames > 1 ? c->slot_frame : nullptr, c->n_group_frames > 1 ? c->n_group_frames : 1u, st);
    return sdn_int::field_forward_f16(c->xyzs, c->dirs, c->live_idx, (const uint32_t *)c->live_counts, c->state, m_bound, c->field_weights,
                                      c->field_bias0, c->grid_table, c->grid_offsets, c->grid_S, c->grid_H, c->bound, c->density_scale,
                                      c->zero_deform, c->sigmas, c->rgbs, expect_points, c->n_group_frames > 1 ? c->slot_frame : nullptr,
                                      c->n_group_frames > 1 ? c->n_group_frames : 1u, st);
}

static bool ctx_ok(const SdnRenderCtx *c) {
    if (!(c && c->rays_o && c->rays_d && c->nears && c->fars && c->alive_a && c->alive_b && c->rays_t && c->weights_sum &&
          c->depth && c->image && c->state && c->live_counts && c->cull_bits))
        return false;
    if (c->n_group_frames > 1) {   // frame group: one occupancy slice per frame, rays split evenly, the per-slot frame scratch
        if (c->n_group_frames > SDN_MAX_GROUP_FRAMES || !c->slot_frame || c->rays_per_frame == 0 ||
            (uint64_t)c->n_group_frames * c->rays_per_frame != c->N)
            return false;
        for (uint32_t f = 0; f < c->n_group_frames; f++)
            if (!c->frame_bitfield[f]) return false;
        return true;
    }
    return c->bitfield != nullptr;
}

int sdn_render_begin(const SdnRenderCtx *c, void *stream) {
    if (!ctx_ok(c)) return SDN_E_BADARG;
    return sdn_int::render_begin(c, nullptr, 0, (hipStream_t)stream);
}

int sdn_render_step_f16(const SdnRenderCtx *c, uint32_t bound_alive, void *stream) {
    return sdn_render_step_f16_ev(c, bound_alive, nullptr, nullptr, stream);
}

int sdn_render_step_f16_ev(const SdnRenderCtx *c, uint32_t bound_alive, void *ev_field_begin, void *ev_field_end, void *stream) {
    if (!c || bound_alive == 0) return SDN_E_BADARG;
    if (bound_alive > c->N) bound_alive = c->N;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *cull = (c->H == 128 && c->C == 1) ? (const uint32_t *)c->cull_bits : nullptr;
    const bool time_march = g_timed_kernel.load() == 1;
    if (time_march && ev_field_begin) (void)hipEventRecord((hipEvent_t)ev_field_begin, st);
    int rc = sdn_int::loop_march(bound_alive, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound, c->dt_gamma, c->max_steps,
                                 c->C, c->H, c->bitfield, c->fars, c->xyzs, c->dirs, c->deltas, cull, c->live_idx,
                                 (uint32_t *)c->live_counts, c->state, sdn_int::frame_sel(c), st, c->rays_tend ? c->sigmas : nullptr);
    if (time_march && ev_field_end) (void)hipEventRecord((hipEvent_t)ev_field_end, st);
    if (time_march) ev_field_begin = ev_field_end = nullptr;
    if (rc) return rc;
    // n_alive * n_step <= N always (n_step <= N / n_alive), and <= 8 * bound_alive
    uint64_t m_bound = (uint64_t)bound_alive * 8u;
    if (m_bound > c->N) m_bound = c->N;
    rc = seal_map(c, (uint32_t)m_bound, st);
    if (rc) return rc;
    if (ev_field_begin) (void)hipEventRecord((hipEvent_t)ev_field_begin, st);
    rc = launch_field(c, (uint32_t)m_bound, 0u, st);
    if (ev_field_end) (void)hipEventRecord((hipEvent_t)ev_field_end, st);
    if (rc) return rc;
    rc = seal_color(c, (uint32_t)m_bound, st);
    if (rc) return rc;
    return sdn_int::loop_composite_compact(bound_alive, c->T_thresh, c->alive_a, c->alive_b, c->rays_t, c->sigmas, c->rgbs, c->deltas,
                                           c->weights_sum, c->depth, c->image, c->state, (uint32_t *)c->block_totals, c->n_out, c->trace,
                                           c->trace + 2 * (size_t)c->n_counters, st);
}

// Whole-frame driver: begin + iterations + finish in one call, so the per-iteration host work is a handful of HIP API
// calls in C (a Python loop costs 100-250 us per iteration on a slow host core, more than the GPU needs for the iteration).
// The loop record's snapshot is copied to pinned host memory on `side_stream` behind an event, so the main stream never
// waits for the host; the host waits for the snapshot of iteration k-1 after it has enqueued iteration k.
//   ev_main[4], ev_copy[4]: caller-created hipEvent_t (no timing needed); host_snap: pinned, 8 ints (4 x {n_alive, call});
//   ev_field: NULL or 2 * max_field_events hipEvent_t with timing enabled, recorded around the fused-field launches
//   (pairs 2k, 2k+1 for iteration k; iterations beyond max_field_events are not timed);
//   iterations_out: number of step calls enqueued (including the trailing no-op one).
}  // extern "C"

namespace {

constexpr uint32_t kRecompactMin = 8192;   // below this many alive rays a launch is latency-bound whatever its list looks like
bool recompact_enabled() {
    static int on = -1;
    if (on < 0) { const char *e = getenv("SDN_RECOMPACT"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
// Re-compact when at most this percentage of the frozen list is still alive.  Measured (profiles/r03_recompaction.txt): a long list
// (a group of 4 full frames freezes ~320 K entries) wants it on nearly every iteration -- 0.431 ms per frame at 95-100 % against 0.459 at
// 50 % and 0.490 without; a short one (one frame, or 8 shards of a frame: ~80 K entries) is best left alone until half of it is dead
// (0.487 / 0.0694 at 50 % against 0.50-0.52 / 0.072-0.076 at 95 %): the four extra launches of a compacting iteration are latency on
// that loop, the waves it saves are proportional to the list.  The cut sits above the longest list one 640 000-ray frame can freeze
// (N / 8 = 80 000).  SDN_RECOMPACT_PCT / SDN_RECOMPACT_MIN override (measurements).
uint32_t recompact_pct(uint32_t list_len) {
    static int v = -2;
    if (v == -2) { const char *e = getenv("SDN_RECOMPACT_PCT"); v = e ? atoi(e) : -1; }
    return v >= 0 ? (uint32_t)v : (list_len > 81920u ? 95u : 50u);
}
uint32_t recompact_min() { static int v = -1; if (v < 0) { const char *e = getenv("SDN_RECOMPACT_MIN"); v = e ? atoi(e) : (int)kRecompactMin; } return (uint32_t)v; }
constexpr int kDriverTimeoutSeconds = 20;   // no iteration of any frame in flight completes for this long: SDN_E_TIMEOUT

// One ray group's loop as a two-phase state machine, so that one host thread can drive several groups round-robin.
struct FrameRun {
    const SdnRenderCtx *c;
    hipStream_t st, side;
    void **ev_main, **ev_copy, **ev_field;
    int32_t *host_snap;
    uint32_t max_field_events;
    uint32_t bound, it;
    bool steady, done;   // steady: see k_composite_march -- two launches per iteration once n_alive <= N / 8
    // Steady mode walks a FROZEN list (dead entries stay in it as -1): once half of it is dead, one iteration ends with the normal
    // mode's compositing + stable compaction instead of the fused kernel and the list is frozen again at its new length -- the
    // marcher's waves are dense again (same rays in the same order, same samples; four more launches on that iteration).
    bool recompact = false;
    uint32_t list_bound = 0;   // upper bound of the frozen list's length
    // Mailbox mode: host_snap is coherent (fine-grained) mapped host memory (sdn_host_mailbox_alloc), the kernels publish each
    // iteration's {tag : n_alive} into it with one 64-bit system-scope store and the host polls -- the main stream then carries
    // nothing but kernels (no event record / stream wait / copy / event wait per iteration).
    void *mail_dev = nullptr;
    uint32_t tag = 0;
    uint32_t last_alive = 0;   // alive rays entering the newest iteration whose count has been read back (a bound for later ones)

    int begin() {
        if (!ctx_ok(c)) return SDN_E_BADARG;
        bound = c->N; it = 0; steady = false; done = false; last_alive = c->N; recompact = false; list_bound = 0;
        mail_dev = nullptr;
        unsigned int flags = 0;
        void *dptr = nullptr;
        if (hipHostGetFlags(&flags, host_snap) == hipSuccess && (flags & hipHostMallocCoherent) &&
            hipHostGetDevicePointer(&dptr, host_snap, 0) == hipSuccess && dptr) {
            static std::atomic<uint32_t> frame_seq{0};
            tag = (frame_seq.fetch_add(1) % 0x7FFFu) + 1u;   // 1 .. 32767: stale publications of an earlier frame never match
            mail_dev = dptr;
        } else {
            (void)hipGetLastError();
        }
        return sdn_int::render_begin(c, mail_dev, tag, st);
    }
    const uint32_t *cull() const { return (c->H == 128 && c->C == 1) ? (const uint32_t *)c->cull_bits : nullptr; }

    // enqueue iteration `it` and the asynchronous read-back of its survivor count
    int enqueue() {
        int32_t *snap_dev = c->trace + 2 * (size_t)c->n_counters;
        void *e0 = (ev_field && it < max_field_events) ? ev_field[2 * it] : nullptr;
        void *e1 = (ev_field && it < max_field_events) ? ev_field[2 * it + 1] : nullptr;
        int rc;
        if (!steady) {
            rc = sdn_render_step_f16_ev(c, bound, e0, e1, st);
        } else {
            uint64_t m_bound = (uint64_t)bound * 8u;
            if (m_bound > c->N) m_bound = c->N;
            rc = seal_map(c, (uint32_t)m_bound, st);
            if (rc) return rc;
            const bool time_march = g_timed_kernel.load() == 1;
            void *m0 = time_march ? e0 : nullptr, *m1 = time_march ? e1 : nullptr;
            if (time_march) e0 = e1 = nullptr;
            if (e0) (void)hipEventRecord((hipEvent_t)e0, st);
            rc = launch_field(c, (uint32_t)m_bound, last_alive * 8u, st);
            if (e1) (void)hipEventRecord((hipEvent_t)e1, st);
            if (!rc) rc = seal_color(c, (uint32_t)m_bound, st);
            if (!rc && m0) (void)hipEventRecord((hipEvent_t)m0, st);
            if (!rc && recompact) {
                // composite this iteration on the frozen list, compact it, freeze the shorter list and march the next iteration on it
                rc = sdn_int::loop_composite_compact(bound, c->T_thresh, c->alive_a, c->alive_b, c->rays_t, c->sigmas, c->rgbs, c->deltas, c->weights_sum,
                                                     c->depth, c->image, c->state, (uint32_t *)c->block_totals, c->n_out, c->trace, snap_dev, st, true);
                const uint32_t nb = last_alive < bound ? last_alive : bound;     // the new list: at most the rays alive when this iteration began
                if (!rc)
                    rc = sdn_int::loop_steady_begin(nb, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound, c->dt_gamma, c->max_steps, c->C,
                                                    c->H, c->bitfield, c->fars, c->xyzs, c->dirs, c->deltas, cull(), c->live_idx, (uint32_t *)c->live_counts,
                                                    c->state, sdn_int::frame_sel(c), st, true);
                bound = list_bound = nb;
                recompact = false;
            } else if (!rc)
                rc = sdn_int::loop_composite_march(bound, c->T_thresh, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound,
                                                   c->dt_gamma, c->max_steps, c->C, c->H, c->bitfield, c->fars, c->sigmas, c->rgbs, c->xyzs,
                                                   c->dirs, c->deltas, c->weights_sum, c->depth, c->image, cull(), c->live_idx,
                                                   (uint32_t *)c->live_counts, c->state, c->n_out, c->trace, snap_dev, sdn_int::frame_sel(c), st);
            if (!rc && m1) (void)hipEventRecord((hipEvent_t)m1, st);
        }
        if (rc) return rc;
        if (mail_dev) return 0;
        const uint32_t slot = it & 3u;
        hipError_t e = hipEventRecord((hipEvent_t)ev_main[slot], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, (hipEvent_t)ev_main[slot], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(host_snap + 2 * slot, snap_dev + 2 * slot, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, side);
        if (e == hipSuccess) e = hipEventRecord((hipEvent_t)ev_copy[slot], side);
        return (int)e;
    }

    // has the survivor count of iteration it-1 arrived (non-blocking)?  settle() will then not wait
    bool ready() const {
        if (it == 0) return true;
        const uint32_t prev = (it - 1) & 3u;
        if (mail_dev) {
            const uint64_t want = ((uint64_t)tag << 16) | it;
            return (__atomic_load_n(reinterpret_cast<const uint64_t *>(host_snap) + prev, __ATOMIC_ACQUIRE) >> 32) == want;
        }
        return hipEventQuery((hipEvent_t)ev_copy[prev]) == hipSuccess;
    }

    // wait for the survivor count of iteration it-1 (iteration `it` is already enqueued), then advance
    int settle() {
        if (it >= 1) {
            const uint32_t prev = (it - 1) & 3u;
            int32_t n_prev;
            if (mail_dev) {
                const uint64_t want = ((uint64_t)tag << 16) | it;     // iteration it-1 publishes call + 1 == it
                const uint64_t *word = reinterpret_cast<const uint64_t *>(host_snap) + prev;
                uint64_t v;
                uint32_t spins = 0;
                auto t0 = std::chrono::steady_clock::now();
                auto t_query = t0;
                while (((v = __atomic_load_n(word, __ATOMIC_ACQUIRE)) >> 32) != want) {
                    if ((++spins & 0x3FFFu) == 0) {
                        // A faulted kernel never publishes: surface the error -- but only after 5 ms without news, and then every 5 ms.
                        // hipStreamQuery puts a marker packet into the stream; asked every 65 536 spins (the cached mailbox word spins at
                        // ~1 ns: once per iteration of a lone frame) it cost 5.9 us in front of the next iteration's first kernel, 58 us of
                        // a 1.04 ms frame (rocprofv3 trace, profiles/r04_lone_frame_timeline.txt)
                        const auto now = std::chrono::steady_clock::now();
                        if (now - t0 > std::chrono::seconds(kDriverTimeoutSeconds)) return SDN_E_TIMEOUT;
                        if (now - t_query > std::chrono::milliseconds(5)) {
                            t_query = now;
                            hipError_t q = hipStreamQuery(st);
                            if (q != hipSuccess && q != hipErrorNotReady) return (int)q;
                        }
                    }
                }
                n_prev = (int32_t)(uint32_t)v;
            } else {
                hipError_t e = hipEventSynchronize((hipEvent_t)ev_copy[prev]);
                if (e != hipSuccess) return (int)e;
                n_prev = host_snap[2 * prev];
            }
            // n_prev: alive rays entering iteration `it`
            if (n_prev <= 0) { done = true; return 0; }
            last_alive = (uint32_t)n_prev;
            if (!steady) {
                bound = (uint32_t)n_prev;
                if ((uint64_t)n_prev * 8u <= c->N) {
                    // iteration `it` (enqueued, normal mode) ends with a compacted list; freeze it and march iteration it+1
                    int rc = sdn_int::loop_steady_begin(bound, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound, c->dt_gamma,
                                                        c->max_steps, c->C, c->H, c->bitfield, c->fars, c->xyzs, c->dirs, c->deltas, cull(),
                                                        c->live_idx, (uint32_t *)c->live_counts, c->state, sdn_int::frame_sel(c), st);
                    if (rc) return rc;
                    steady = true;
                    list_bound = bound;
                }
            } else if (recompact_enabled() && (uint64_t)n_prev * 100u <= (uint64_t)list_bound * recompact_pct(list_bound) && (uint32_t)n_prev >= recompact_min()) {
                recompact = true;     // the iteration enqueued next ends with the compaction
            }
        }
        it++;
        if (it > c->max_steps + 1) done = true;
        return 0;
    }
};

}  // namespace

extern "C" {

int sdn_render_frame_f16(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream, void *side_stream,
                         void **ev_main, void **ev_copy, int32_t *host_snap, void **ev_field, uint32_t max_field_events,
                         uint32_t *iterations_out) {
    if (!c || !image_out || !depth_out || !side_stream || !ev_main || !ev_copy || !host_snap) return SDN_E_BADARG;
    FrameRun r{c, (hipStream_t)stream, (hipStream_t)side_stream, ev_main, ev_copy, ev_field, host_snap, max_field_events, 0, 0, false, false};
    int rc = r.begin();
    while (!rc && !r.done) {
        rc = r.enqueue();
        if (!rc) rc = r.settle();
    }
    if (rc) return rc;
    if (iterations_out) *iterations_out = r.it + 1;
    return sdn_render_finish(c, bg_color, image_out, depth_out, stream);
}

// A stream of frames (camera path / time steps) through n_ctx loop contexts used in turn, each on its own stream: the next frame
// begins as soon as a context is free and the alive rays of the newest frame in flight have dropped to N / overlap_div
// (1 = at once), so that the latency-bound parts of one frame -- marching chains, the tail iterations with a few thousand
// rays, launch gaps -- run under the throughput-bound field kernels of another.  Frames are independent (separate state, samples and outputs per context): every
// frame is bit-identical to the one `sdn_render_frame_f16` renders; only throughput changes.  ctxs[n_ctx] carry everything but the
// per-frame pointers, which come from rays_o / rays_d / image_outs / depth_outs [n_frames] (entries may repeat); host_snap: 8 ints
// per context; ev_main / ev_copy: 4 events per context (event + copy read-back only); ev_field_frames: NULL or [n_frames] pointers
// (NULL entries allowed) to 2 * max_field_events timing events recorded around that frame's field launches; exclusive_frames: NULL or
// [n_frames] flags -- a flagged frame starts only when no other frame is in flight and nothing starts until it is done.
int sdn_render_frames_pipelined_f16(const SdnRenderCtx *const *ctxs, uint32_t n_ctx, uint32_t n_frames, const float *const *rays_o,
                                    const float *const *rays_d, float *const *image_outs, float *const *depth_outs, float bg_color,
                                    uint32_t overlap_div, void *const *streams, void *const *side_streams, void **ev_main, void **ev_copy,
                                    int32_t *host_snap, void *const *ev_field_frames, uint32_t max_field_events, const uint8_t *exclusive_frames,
                                    const SdnFrameTime *frame_times, void *const *done_events, uint32_t *iterations_out) {
    constexpr uint32_t kMaxCtx = 8;
    if (!ctxs || n_ctx == 0 || n_ctx > kMaxCtx || !rays_o || !rays_d || !image_outs || !depth_outs || !streams || !side_streams ||
        !ev_main || !ev_copy || !host_snap)
        return SDN_E_BADARG;
    for (uint32_t s = 0; s < n_ctx; s++)
        if (!ctxs[s] || !ctxs[s]->aabb || !side_streams[s]) return SDN_E_BADARG;
    if (n_frames == 0) return 0;
    if (overlap_div == 0) overlap_div = 1;
    SdnRenderCtx local[kMaxCtx];
    FrameRun runs[kMaxCtx];
    int frame_of[kMaxCtx];             // frame a context is rendering, -1 = free
    for (uint32_t s = 0; s < n_ctx; s++) { local[s] = *ctxs[s]; frame_of[s] = -1; }
    uint32_t next = 0, finished = 0;
    auto start = [&](uint32_t slot) -> int {
        if (!rays_o[next] || !rays_d[next] || !image_outs[next] || !depth_outs[next]) return SDN_E_BADARG;
        local[slot].rays_o = rays_o[next];
        local[slot].rays_d = rays_d[next];
        if (frame_times) {   // this frame's (frame group's) own time: occupancy slice(s), time-encoding bias, canonical-frame flag(s)
            const SdnFrameTime &ft = frame_times[next];
            if (!ft.field_bias0 || !ft.bitfield[0]) return SDN_E_BADARG;
            local[slot].bitfield = ft.bitfield[0];
            for (uint32_t f = 0; f < SDN_MAX_GROUP_FRAMES; f++) local[slot].frame_bitfield[f] = ft.bitfield[f];
            for (uint32_t f = 0; f < SDN_MAX_GROUP_FRAMES; f++) local[slot].frame_cull[f] = ft.cull_grid[f];
            local[slot].field_bias0 = ft.field_bias0;
            local[slot].zero_deform = (int32_t)ft.zero_deform;
        }
        void **evf = ev_field_frames ? (void **)ev_field_frames[next] : nullptr;   // timing events of this frame's field launches, or none
        runs[slot] = FrameRun{&local[slot], (hipStream_t)streams[slot], (hipStream_t)side_streams[slot], ev_main + 4 * slot, ev_copy + 4 * slot,
                              evf, host_snap + 8 * slot, evf ? max_field_events : 0u, 0, 0, false, false};
        frame_of[slot] = (int)next++;
        return runs[slot].begin();
    };
    // Event-driven: every frame in flight always has exactly one iteration enqueued beyond the one whose survivor count the host
    // has seen; whichever frame's count arrives is advanced at once (oldest first when several are ready), so a frame never
    // idles behind another frame's longer iteration.
    const bool stats = getenv("SDN_DRIVER_STATS") != nullptr;   // host-side time split of the driver loop, to stderr
    double t_work = 0, t_idle = 0;
    uint32_t n_loops = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto launch_next = [&]() -> int {   // start the next frame if a context is free and the overlap rule allows it
        if (next >= n_frames) return 0;
        const uint32_t slot = next % n_ctx;
        if (frame_of[slot] >= 0) return 0;
        int newest = -1;
        bool any_active = false, excl_in_flight = false;
        for (uint32_t c = 0; c < n_ctx; c++) {
            if (frame_of[c] < 0) continue;
            any_active = true;
            if (newest < 0 || frame_of[c] > frame_of[newest]) newest = (int)c;
            if (exclusive_frames && exclusive_frames[frame_of[c]]) excl_in_flight = true;
        }
        bool ok = !any_active || (uint64_t)runs[newest].last_alive * overlap_div <= local[slot].N;
        // an exclusive frame has the device to itself (used to time its kernels undisturbed): it starts only when nothing
        // else is in flight, and nothing starts while it is
        if (exclusive_frames && any_active && (exclusive_frames[next] || excl_in_flight)) ok = false;
        if (!ok) return 0;
        int r = start(slot);
        if (!r) r = runs[slot].enqueue();
        return r;
    };
    // every error return first waits for all streams: the caller must never reuse buffers that kernels still write
    auto fail = [&](int code) -> int {
        for (uint32_t c = 0; c < n_ctx; c++) { (void)hipStreamSynchronize((hipStream_t)streams[c]); (void)hipStreamSynchronize((hipStream_t)side_streams[c]); }
        (void)hipGetLastError();
        return code;
    };
    int rc = launch_next();
    if (rc) return fail(rc);
    auto t_last = stats ? now() : 0.0;
    uint32_t idle_spins = 0;
    auto t_progress = std::chrono::steady_clock::now();
    auto t_query = t_progress;
    while (finished < n_frames) {
        n_loops++;
        bool progress = false;
        // contexts in frame order, oldest first
        uint32_t order[kMaxCtx], n_act = 0;
        for (uint32_t s2 = 0; s2 < n_ctx; s2++)
            if (frame_of[s2] >= 0) order[n_act++] = s2;
        for (uint32_t a = 1; a < n_act; a++)
            for (uint32_t b = a; b > 0 && frame_of[order[b]] < frame_of[order[b - 1]]; b--) { const uint32_t t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
        for (uint32_t a = 0; a < n_act; a++) {
            const uint32_t s2 = order[a];
            if (!runs[s2].ready()) continue;
            progress = true;
            rc = runs[s2].settle();
            if (rc) return fail(rc);
            if (runs[s2].done) {
                const int f = frame_of[s2];
                rc = sdn_render_finish(&local[s2], bg_color, image_outs[f], depth_outs[f], streams[s2]);
                if (rc) return fail(rc);
                if (done_events && done_events[f]) {
                    hipError_t e = hipEventRecord((hipEvent_t)done_events[f], (hipStream_t)streams[s2]);
                    if (e != hipSuccess) return fail((int)e);
                }
                if (iterations_out) __atomic_store_n(&iterations_out[f], runs[s2].it + 1, __ATOMIC_RELEASE);
                frame_of[s2] = -1;
                finished++;
            } else {
                rc = runs[s2].enqueue();
                if (rc) return fail(rc);
            }
            rc = launch_next();
            if (rc) return fail(rc);
        }
        if (!progress) {
            rc = launch_next();
            if (rc) return fail(rc);
            if ((++idle_spins & 0x3FFFu) == 0) {
                // nothing has arrived for 5 ms (and every 5 ms from then on): surface a device error, and give up after 20 s without any
                // iteration completing (a hung kernel never publishes its mailbox word and hipStreamQuery keeps answering "not ready").
                // Time-based, not spin-count-based: every hipStreamQuery is a marker packet in that loop's stream (see FrameRun::settle)
                const auto now_c = std::chrono::steady_clock::now();
                if (now_c - t_progress > std::chrono::milliseconds(5) && now_c - t_query > std::chrono::milliseconds(5)) {
                    t_query = now_c;
                    for (uint32_t c = 0; c < n_ctx; c++)
                        if (frame_of[c] >= 0) { hipError_t q = hipStreamQuery((hipStream_t)streams[c]); if (q != hipSuccess && q != hipErrorNotReady) return fail((int)q); }
                }
                if (now_c - t_progress > std::chrono::seconds(kDriverTimeoutSeconds)) return SDN_E_TIMEOUT;
            }
        } else {
            idle_spins = 0;
            t_progress = std::chrono::steady_clock::now();
        }
        if (stats) { const double t = now(); (progress ? t_work : t_idle) += t - t_last; t_last = t; }
    }
    if (stats)
        fprintf(stderr, "[sdn driver] frames %u contexts %u loops %u  host busy %.0f us  idle-polling %.0f us  mailbox %d\n", n_frames, n_ctx, n_loops, t_work,
                t_idle, runs[0].mail_dev != nullptr);
    return 0;
}

// 32 bytes (4 snapshot slots) per ray group of coherent, device-mapped host memory for the frame drivers' read-back.
void *sdn_host_mailbox_alloc(uint32_t groups) {
    void *p = nullptr;
    if (groups == 0 || hipHostMalloc(&p, 32u * groups, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(p, 0, 32u * groups);
    return p;
}

int sdn_host_mailbox_free(void *p) { return p ? (int)hipHostFree(p) : 0; }

int sdn_render_finish(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream) {
    if (!c || !image_out || !depth_out) return SDN_E_BADARG;
    return sdn_int::loop_finish(c->N, c->nears, c->fars, c->weights_sum, c->depth, c->image, bg_color, image_out, depth_out,
                                (hipStream_t)stream);
}

}  // extern "C"
