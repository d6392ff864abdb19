// Device-driven inference loop for one frame (dnerf/renderer.py:333-381 of the reference, `-O` numerics).
//
// The reference re-enters Python after every loop iteration: `rays_alive[rays_alive >= 0]` needs the survivor count on
// the host (a device->host sync), and `n_step` is derived from it.  Here n_alive, n_step, the step budget, the
// ping-pong side of the alive list and the per-iteration live-sample counter live in a 32-byte device record that
// the kernels of an iteration read and a one-thread kernel advances, so the host can enqueue iteration k+1 while
// iteration k runs; it only needs an UPPER BOUND of n_alive to size the grids (the survivor count it read back one
// iteration earlier), and learns that the loop is over one iteration late (that iteration is a no-op).
// The schedule -- n_step = clamp(N // n_alive, 1, 8), stop at max_steps -- and every sample are the reference's.
#include "sdn_common.h"
#include "sdn_internal.h"

extern "C" {

int sdn_render_finish(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream);
int sdn_render_step_f16_ev(const SdnRenderCtx *c, uint32_t bound_alive, void *ev_field_begin, void *ev_field_end, void *stream);

int sdn_render_begin(const SdnRenderCtx *c, void *stream) {
    if (!c || !c->rays_o || !c->rays_d || !c->nears || !c->fars || !c->bitfield || !c->alive_a || !c->alive_b || !c->rays_t ||
        !c->weights_sum || !c->depth || !c->image || !c->state || !c->live_counts || !c->cull_bits)
        return SDN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int rc = sdn_int::loop_begin(c->N, c->max_steps, c->nears, c->alive_a, c->rays_t, c->weights_sum, c->depth, c->image, c->state,
                                 c->live_counts, c->n_counters, st);
    if (rc) return rc;
    if (c->H == 128 && c->C == 1) rc = sdn_int::build_cull(c->bitfield, (uint32_t *)c->cull_bits, st);
    return rc;
}

int sdn_render_step_f16(const SdnRenderCtx *c, uint32_t bound_alive, void *stream) {
    return sdn_render_step_f16_ev(c, bound_alive, nullptr, nullptr, stream);
}

int sdn_render_step_f16_ev(const SdnRenderCtx *c, uint32_t bound_alive, void *ev_field_begin, void *ev_field_end, void *stream) {
    if (!c || bound_alive == 0) return SDN_E_BADARG;
    if (bound_alive > c->N) bound_alive = c->N;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *cull = (c->H == 128 && c->C == 1) ? (const uint32_t *)c->cull_bits : nullptr;
    int rc = sdn_int::loop_march(bound_alive, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound, c->dt_gamma, c->max_steps,
                                 c->C, c->H, c->bitfield, c->fars, c->xyzs, c->dirs, c->deltas, cull, c->live_idx,
                                 (uint32_t *)c->live_counts, c->state, st);
    if (rc) return rc;
    // n_alive * n_step <= N always (n_step <= N / n_alive), and <= 8 * bound_alive
    uint64_t m_bound = (uint64_t)bound_alive * 8u;
    if (m_bound > c->N) m_bound = c->N;
    if (ev_field_begin) (void)hipEventRecord((hipEvent_t)ev_field_begin, st);
    rc = sdn_int::field_forward_f16(c->xyzs, c->dirs, c->live_idx, (const uint32_t *)c->live_counts, c->state, (uint32_t)m_bound,
                                    c->field_weights, c->field_bias0, c->grid_table, c->grid_offsets, c->grid_S, c->grid_H, c->bound,
                                    c->density_scale, c->zero_deform, c->sigmas, c->rgbs, st);
    if (ev_field_end) (void)hipEventRecord((hipEvent_t)ev_field_end, st);
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
int sdn_render_frame_f16(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream, void *side_stream,
                         void **ev_main, void **ev_copy, int32_t *host_snap, void **ev_field, uint32_t max_field_events,
                         uint32_t *iterations_out) {
    if (!c || !image_out || !depth_out || !side_stream || !ev_main || !ev_copy || !host_snap) return SDN_E_BADARG;
    hipStream_t st = (hipStream_t)stream, side = (hipStream_t)side_stream;
    int rc = sdn_render_begin(c, stream);
    if (rc) return rc;
    int32_t *snap_dev = c->trace + 2 * (size_t)c->n_counters;
    const uint32_t *cull = (c->H == 128 && c->C == 1) ? (const uint32_t *)c->cull_bits : nullptr;
    uint32_t bound = c->N, it = 0;
    bool steady = false;  // see k_composite_march: two launches per iteration once n_alive <= N / 8
    for (;;) {
        void *e0 = (ev_field && it < max_field_events) ? ev_field[2 * it] : nullptr;
        void *e1 = (ev_field && it < max_field_events) ? ev_field[2 * it + 1] : nullptr;
        if (!steady) {
            rc = sdn_render_step_f16_ev(c, bound, e0, e1, stream);
        } else {
            uint64_t m_bound = (uint64_t)bound * 8u;
            if (m_bound > c->N) m_bound = c->N;
            if (e0) (void)hipEventRecord((hipEvent_t)e0, st);
            rc = sdn_int::field_forward_f16(c->xyzs, c->dirs, c->live_idx, (const uint32_t *)c->live_counts, c->state, (uint32_t)m_bound,
                                            c->field_weights, c->field_bias0, c->grid_table, c->grid_offsets, c->grid_S, c->grid_H, c->bound,
                                            c->density_scale, c->zero_deform, c->sigmas, c->rgbs, st);
            if (e1) (void)hipEventRecord((hipEvent_t)e1, st);
            if (!rc)
                rc = sdn_int::loop_composite_march(bound, c->T_thresh, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound,
                                                   c->dt_gamma, c->max_steps, c->C, c->H, c->bitfield, c->fars, c->sigmas, c->rgbs, c->xyzs,
                                                   c->dirs, c->deltas, c->weights_sum, c->depth, c->image, cull, c->live_idx,
                                                   (uint32_t *)c->live_counts, c->state, c->n_out, c->trace, snap_dev, st);
        }
        if (rc) return rc;
        const uint32_t slot = it & 3u;
        hipError_t e = hipEventRecord((hipEvent_t)ev_main[slot], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, (hipEvent_t)ev_main[slot], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(host_snap + 2 * slot, snap_dev + 2 * slot, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, side);
        if (e == hipSuccess) e = hipEventRecord((hipEvent_t)ev_copy[slot], side);
        if (e != hipSuccess) return (int)e;
        if (it >= 1) {
            const uint32_t prev = (it - 1) & 3u;
            e = hipEventSynchronize((hipEvent_t)ev_copy[prev]);
            if (e != hipSuccess) return (int)e;
            const int32_t n_prev = host_snap[2 * prev];  // alive rays entering iteration `it` (already enqueued)
            if (n_prev <= 0) break;
            if (!steady) {
                bound = (uint32_t)n_prev;
                if ((uint64_t)n_prev * 8u <= c->N) {
                    // iteration `it` (enqueued above, normal mode) ends with a compacted list; freeze it and march iteration it+1
                    rc = sdn_int::loop_steady_begin(bound, c->alive_a, c->alive_b, c->rays_t, c->rays_o, c->rays_d, c->bound, c->dt_gamma,
                                                    c->max_steps, c->C, c->H, c->bitfield, c->fars, c->xyzs, c->dirs, c->deltas, cull,
                                                    c->live_idx, (uint32_t *)c->live_counts, c->state, st);
                    if (rc) return rc;
                    steady = true;
                }
            }
        }
        it++;
        if (it > c->max_steps + 1) break;
    }
    if (iterations_out) *iterations_out = it + 1;
    return sdn_render_finish(c, bg_color, image_out, depth_out, stream);
}

int sdn_render_finish(const SdnRenderCtx *c, float bg_color, float *image_out, float *depth_out, void *stream) {
    if (!c || !image_out || !depth_out) return SDN_E_BADARG;
    return sdn_int::loop_finish(c->N, c->nears, c->fars, c->weights_sum, c->depth, c->image, bg_color, image_out, depth_out,
                                (hipStream_t)stream);
}

}  // extern "C"
