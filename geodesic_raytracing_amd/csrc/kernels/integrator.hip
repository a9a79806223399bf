// ------------------------------------------------------------------------------------------------
// the integrator (cl.cl:3273-3346, 3400-3456, 3954-4247)

#ifdef ADAPTIVE_PRECISION
#define GR_W_MAX ((float)((W_V1 > W_V2 ? W_V1 : W_V2) > (W_V3 > W_V4 ? W_V3 : W_V4) ? (W_V1 > W_V2 ? W_V1 : W_V2) : (W_V3 > W_V4 ? W_V3 : W_V4)))

// returns diff, writes the unclamped step suggestion (acceleration_to_precision)
__device__ __forceinline__ float acceleration_to_precision(float4 acc, float max_acceleration, float& next_ds) {
    float4 wa = f4(acc.x * (float)(W_V1), acc.y * (float)(W_V2), acc.z * (float)(W_V3), acc.w * (float)(W_V4));
    float current = __builtin_sqrtf(dot4(wa, wa)) * 0.01f;
    current /= GR_W_MAX;
    const float scale = 65536.f;                 // I_HATE_COMPUTERS
    float err = max_acceleration;
    float diff = current * scale;
    float floor_diff = err * scale / 1e10f;      // pow(max_timestep = 100000, 2)
    if (diff < floor_diff) diff = floor_diff;
    next_ds = __builtin_sqrtf(err * scale) * __builtin_amdgcn_rsqf(diff);   // sqrt((err * scale) / diff); first factor is loop-invariant
    return diff;
}
#endif

enum { DS_NONE = 0, DS_SKIP = 1, DS_RETURN = 2 };

struct ray_state {
    float4 position, velocity, acceleration;
    float next_ds;
    float running_dlambda_dnew;
    float f_in_x;
    // progress of a ray that is integrated in several visits (ray compaction): accepted steps, attempts
    int steps;
    unsigned int tries;
};

// outcome of integrating one ray
enum { RAY_LOST = 0, RAY_TERMINATED = 1 };

// Integrates until termination.  Returns RAY_TERMINATED when the ray reached the outer boundary (or the
// SINGULAR terminator) - position/velocity/running_dlambda_dnew are then final - and RAY_LOST on any
// early return of the reference (singularity guards, NaN, step cap), where nothing is written back.
//
// RESUMABLE (ray compaction): the loop also stops - `paused` - as soon as fewer than keep_lanes lanes of the wave are still
// iterating, with everything it carries between iterations saved in `s`, so that the caller can hand the idle lanes new rays
// and call again; integrate_begin prepares `s` for the first visit.  The arithmetic of a ray does not depend on the visits.
__device__ __forceinline__ void integrate_begin(ray_state& s, dfg_t dfg) {
    s.f_in_x = __builtin_fabsf(s.velocity.x);
    s.next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    (void)acceleration_to_precision(s.acceleration, GET_FEATURE(max_acceleration_change, dfg), s.next_ds);
#endif
    s.running_dlambda_dnew = 1;
    s.steps = 0;
    s.tries = 0;
}

// ---- the integrator as it runs by default ---------------------------------------------------------------------------------
// The same algorithm written for the machine's costs.  Measured facts it is built on (MI355X, 4K Kerr): one more VALU
// instruction per attempt costs ~0.35 % of the frame, a scalar one ~0.15 %; a step is REJECTED once in ~35 000 attempts
// (oracle count, 192x108 Kerr: 357 of 12.7 M), so everything is arranged for the accepting path and a rejection may be slow.
//   * two attempts per trip, the state ping-ponging between two register sets: the accepted state is written where the next
//     attempt reads it, and only a rejection copies (the one-attempt loop paid 7 v_mov / v_xor per attempt to move the new
//     state into the loop-carried registers);
//   * the controller works on the squared, scaled error q = (|W a| 0.01 / Wmax 65536)^2: one multiply, one max against the
//     squared floor, suggestion = sqrt(err 65536) q^(-1/4) (v_sqrt, v_rsq), clamp by v_med3; the singularity test
//     (cl.cl:3446-3449) is one compare of q in the hot path, its second condition only behind it;
//   * IS_DEGENERATE (cl.cl:4235-4244) on the new velocity alone inside the loop - a non-finite acceleration makes the velocity
//     computed from it non-finite in the same step, a non-finite position needs a non-finite velocity first - and on all three
//     vectors once after the loop, where the outcome is decided;
//   * the step cap (cl.cl:3974: 16384 accepted steps) is the borrow of the subtraction that counts the steps down; the
//     attempt count the profiling launches ask for follows from it after the loop;
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// dst = src as instructions the compiler can neither turn into selects nor move out of the (rarely executed) block they are in
__device__ __forceinline__ void overwrite(float4& dst, float4 src) {
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "+v"(dst.x), "+v"(dst.y), "+v"(dst.z), "+v"(dst.w) : "v"(src.x), "v"(src.y), "v"(src.z), "v"(src.w));
}

// the same with a wave-uniform first operand (a feature value, a literal): no VGPR is spent on it
__device__ __forceinline__ float min_f32_uniform(float uniform, float x) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(x)); return r; }
__device__ __forceinline__ float max_f32_uniform(float uniform, float x) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(x)); return r; }

#ifdef ADAPTIVE_PRECISION
struct step_controller {
    float floor_q, root, singular_q, min_step;
    float root99, floor_step, ambient;   // 0.99 root; min(min_step, ambient); the largest near step (0.2)
    __device__ __forceinline__ step_controller(float max_acceleration, float min_step_in, float ambient_in) {
        const float scale = 65536.f;
        const float floor_diff = max_acceleration * scale / 1e10f;
        floor_q = floor_diff * floor_diff;
        root = __builtin_sqrtf(max_acceleration * scale);
        root99 = 0.99f * root;
        const float singular = max_acceleration * 10000 * scale;
        singular_q = singular * singular;
        min_step = min_step_in;
        ambient = ambient_in;
        floor_step = __builtin_fminf(min_step_in, ambient_in);
    }
    // diff^2 of acceleration_to_precision (cl.cl:3400-3429) before its floor.  A NaN component makes it NaN, an infinite one infinite.
    __device__ __forceinline__ float raw_error_q(float4 acc) const {
        const float k = 0.01f * 65536.f / GR_W_MAX;
        const float wx = acc.x * (float)(W_V1), wy = acc.y * (float)(W_V2), wz = acc.z * (float)(W_V3), ww = acc.w * (float)(W_V4);
        const float d2 = __builtin_fmaf(wx, wx, __builtin_fmaf(wy, wy, __builtin_fmaf(wz, wz, ww * ww)));
        return d2 * (k * k);
    }
    // ... floored (the ray's first step suggestion, where nothing clamps the result)
    __device__ __forceinline__ float error_q(float4 acc) const { return max_f32_uniform(floor_q, raw_error_q(acc)); }
    __device__ __forceinline__ float suggestion(float q) const { return root * __builtin_amdgcn_rsqf(__builtin_sqrtf(q)); }
    // every weight non-zero: a NaN anywhere in an acceleration shows in its error (the loop's degeneracy test rests on it)
    static constexpr bool error_sees_every_component = (float)(W_V1) != 0 && (float)(W_V2) != 0 && (float)(W_V3) != 0 && (float)(W_V4) != 0;
};
#endif

template <bool LIBM> struct trig_flavour { static constexpr bool value = LIBM; };

//
// PARKABLE (parking, trace.hip): the plain loop - a ray starts exactly as it does there - that can also be left, once per trip (two
// attempts) and not before park_trips trips of this visit, when fewer than keep_lanes lanes are still iterating; `resumed` (wave-uniform)
// says that `s` holds a ray left that way, which then goes on as if it had never stopped.
template <bool RESUMABLE, bool PARKABLE = false>
__device__ __forceinline__ int integrate_pingpong(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts, int keep_lanes, bool& paused,
                                                  bool resumed = false, unsigned int park_trips = 0) {
    float4 p0 = s.position, v0 = s.velocity, a0 = s.acceleration;
    float4 p1 = p0, v1 = v0, a1 = a0;
    const bool carried = RESUMABLE || (PARKABLE && resumed);   // what the loop carries between attempts comes from `s`
    if (PARKABLE && !resumed) { s.steps = 0; s.tries = 0; }
    const float f_in_x = carried ? s.f_in_x : __builtin_fabsf(v0.x);
    if (PARKABLE) s.f_in_x = f_in_x;
    const float subambient_precision = 0.5f;
    const float ambient_precision = 0.2f;
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    const step_controller controller(GET_FEATURE(max_acceleration_change, dfg), GET_FEATURE(min_step, dfg), ambient_precision);
    if (carried) next_ds = s.next_ds;
    else {
        float4 a = a0;
#ifdef IS_CONSTANT_THETA
        a.z = 0;
#endif
        next_ds = controller.suggestion(controller.error_q(a));
    }
    // The loop carries the step suggestion as the step a ray inside the precision radius would take: min(suggestion, ambient), the only
    // way the reference ever reads it (cl.cl:4101-4107).  Every update below leaves it clamped the same way.
    next_ds = min_f32_uniform(ambient_precision, next_ds);
#endif
    const float new_max = GET_FEATURE(max_precision_radius, dfg);
    const float new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    float running = carried ? s.running_dlambda_dnew : 1.f;
    const int loop_limit = 4096 * 4;
    // accepted steps the ray may still take (cl.cl:3974: 16384 in all).  Every attempt() entered takes one - the borrow of that very
    // subtraction is the step-cap test - and a rejection (rare) gives it back.
    const unsigned int budget_before = (unsigned int)(loop_limit - (carried ? s.steps : 0));
    unsigned int budget = budget_before;
    unsigned int rejections = 0;
    paused = false;

    auto stop_lost = [&](float4 pos, float4 vel, float4 acc, float run) {
        bool lost = false;
#ifdef HAS_CYLINDRICAL_SINGULARITY
        lost |= pos.y < CYLINDRICAL_TERMINATOR;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        // |dt/dlambda| > 1000 + its initial value AND |d2t/dlambda2| > 100 (cl.cl:4118-4127): the first condition alone in the hot path,
        // the second only in a wave one of whose rays meets the first (a compare is two issue slots, the wave-uniform branch a scalar one)
        bool runaway = __builtin_fabsf(vel.x / run) > 1000 + f_in_x;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(runaway) != 0, 0)) {
            asm volatile("; second condition of the runaway test");
            runaway = runaway && __builtin_fabsf(acc.x / run) > 100;
        }
        lost |= runaway;
#endif
        (void)pos; (void)vel; (void)acc; (void)run;
        return lost;
    };
    // |r| >= universe_size, and for a SINGULAR metric |r| < its terminator (cl.cl:4086-4095).  Where r is a square root (a Cartesian
    // chart: GR_POLAR_R_SQUARED) the comparison is made on its argument: sqrt is monotonic, the two forms differ by which side a
    // radius within an ulp of the bound falls on.
#ifdef GR_POLAR_R_SQUARED
    const float universe2 = universe * universe;
    auto stop_terminated = [&](float4 position, float4 polar) {
        (void)polar;
        const float r2 = gm::polar_radius_squared(position, cfg);
        bool t = r2 >= universe2;
#ifdef SINGULAR
        t |= r2 < (float)(SINGULAR_TERMINATOR) * (float)(SINGULAR_TERMINATOR);
#endif
        return t;
    };
#else
    auto stop_terminated = [&](float4 position, float4 polar) {
        (void)position;
        bool t = __builtin_fabsf(polar.y) >= universe;
#ifdef SINGULAR
        t |= __builtin_fabsf(polar.y) < SINGULAR_TERMINATOR;
#endif
        return t;
    };
#endif
    const float far_offset = ambient_precision - 0.1f * new_max;
    // One Verlet attempt from (p, v, a): the state the next attempt starts from goes to (po, vo, ao) - the new state, or the old
    // one again after a rejection.  Returns true when the loop is to be left: the ray is done and (p, v, a) is its final state
    // (what the loop carries - step suggestion, budget - is then that of the abandoned attempt); or pause_wave was raised
    // (RESUMABLE): the step was taken, the wave wants new rays, the state is (po, vo, ao).
    bool pause_wave = false;
    auto attempt = [&](auto libm, float4 position, float4 velocity, float4 acceleration, float4& po, float4& vo, float4& ao, float& ds_used,
                       float& running_before) -> bool {
        // the for-loop condition of the reference, then its loop-top exits (cl.cl:3974, 4086-4130)
        if (__builtin_expect(__builtin_usub_overflow(budget, 1u, &budget), 0)) return true;
#ifdef IS_CONSTANT_THETA
        position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
#if defined(GR_DISTANCE_SQUARED_OF_GENERIC) && defined(ADAPTIVE_PRECISION)
        // the distance is a square root of the position (a Cartesian chart): inside <=> its argument < radius^2, and the root - only
        // the far step reads it - is taken in a wave that has a ray outside (a v_sqrt_f32 is four multiplies' issue time)
        const float ar2 = gm::distance_squared_from(position, cfg);
        const bool inside = ar2 < new_max * new_max;
        const unsigned long long inside_lanes = __builtin_amdgcn_ballot_w64(inside);
        const bool wave_inside = inside_lanes != 0;
        const float near_ds = next_ds;
        float far_ds = near_ds;
        if (inside_lanes != __builtin_amdgcn_ballot_w64(true)) {
            asm volatile("; a ray outside the precision radius: the far step");   // (keeps the root from being hoisted out of the branch)
            far_ds = __builtin_fmaf(0.1f, __builtin_sqrtf(ar2), far_offset);
        }
#else
        const float ar = __builtin_fabsf(gm::distance_to_object_from(position, polar, cfg));
        const bool inside = ar < new_max;
        const bool wave_inside = __builtin_amdgcn_ballot_w64(inside) != 0;   // (taken here, where the compare is: one scalar instruction)
        (void)wave_inside;
#ifdef ADAPTIVE_PRECISION
        const float near_ds = next_ds;   // carried clamped (above, and every update below)
#else
        const float near_ds = min_f32_uniform(ambient_precision, mixf(ambient_precision, subambient_precision, (clampf(ar, new_min, new_max) - new_min) / (new_max - new_min)));
#endif
        const float far_ds = __builtin_fmaf(0.1f, ar, far_offset);   // 0.1 (|r| - max_precision_radius) + ambient
#endif
        const float ds = inside ? near_ds : far_ds;
        ds_used = ds;
        running_before = running;
        if (stop_lost(position, velocity, acceleration, running) | stop_terminated(position, polar)) return true;
        GR_PROBE_EXTRA_INSTRUCTIONS(ds)
        // velocity Verlet (step_verlet, cl.cl:3273-3346) in the reference's operation order.  (Through the half-kicked velocity
        // h = v + a ds/2 - x' = x + h ds, v~ = h + a ds/2, v' = h + a' ds/2 - it is 16 fmas instead of 20 operations and the same
        // algebra, but its roundings are not the reference's: measured against the golden pixels the RMSE of the Kerr cases went
        // from 7e-6 to 3.7e-5 and pixels off by > 1e-3 from 0-1 to 3-6 per fixture.  Parity first: 4 instructions.)
        const float half_ds = 0.5f * ds, half_ds2 = half_ds * ds;
        const float4 next_position = position + velocity * ds + acceleration * half_ds2;
        const float4 predicted = velocity + acceleration * ds;
        float4 next_acceleration = GR_PROBE_ACCELERATION(gm::geodesic_acceleration_with<decltype(libm)::value>(next_position, predicted, cfg), acceleration, predicted, next_position);
        // (The acceleration above is 135 and more vector instructions in a row.  A wave that issues such a stretch back to back
        // leaves the SIMD's vector port idle part of the time; the host's pass over the compiled code - csrc/codeobject.cpp,
        // break_vector_runs - puts an s_nop after every 8th vector instruction of a run, which is worth 25 % here.)
        float4 next_velocity = velocity + (acceleration + next_acceleration) * half_ds;
        if (reparam) {
            const float md = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x), __builtin_fabsf(next_velocity.y)),
                                             __builtin_fmaxf(__builtin_fabsf(next_velocity.z), __builtin_fabsf(next_velocity.w)));
            const float K = 1 / md;
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
            running *= K;   // also on an attempt that is then rejected, as the reference does (cl.cl:4152-4154)
        }

        // The new state is written where the next attempt reads it; a rejection (once in ~35 000 attempts) puts the old state
        // back over it.  The copies are opaque to the compiler on purpose: as plain assignments it turns the two outcomes into
        // twelve selects per attempt.
        po = next_position;
        vo = next_velocity;
        ao = next_acceleration;
        bool dead = false;
#ifdef ADAPTIVE_PRECISION
        // calculate_ds_error (cl.cl:3431-3456), then IS_DEGENERATE on an accepted step (cl.cl:4235-4244) - arranged so that the path every
        // attempt takes holds TWO compares, everything that happens once in thousands of attempts sits behind one wave-uniform branch,
        // and no lane mask is ever turned into a register and back (round 5; the loop had eight compares, five selects and four
        // min / max here and at its top - instructions that issue at half the rate of a multiply):
        //   * the controller's arithmetic runs on every lane (a wave issues it for all lanes anyway) on the carried near step, which IS
        //     ds for a ray inside the precision radius; only the update of the carried step is selected by `inside`;
        //   * 0.99 ds clamp(suggested / ds, 0.3, 2), then max(., min_step), then the min(., ambient) of the next attempt's loop top:
        //     two v_med3 (the second does both one-sided clamps; floor_step = min(min_step, ambient) keeps it a median);
        //   * the floor of the error (cl.cl:3414-3419) is gone from the loop: it only bounds the suggestion from above, by 1e5, and the
        //     clamp to 1.98 ds <= 0.4 follows;
        //   * a NaN acceleration - a singularity overshot, or a sin / cos argument outside the polynomial's range (metric.hip: the
        //     poison) - makes the error NaN, and fma(error, 0, step) hands that to the one compare that also finds a rejected step:
        //     !(step >= ds / 1.95).  (IS_DEGENERATE's test of the velocity for infinities without a NaN finds them one attempt later,
        //     as NaNs; such a ray is lost either way and writes nothing.)
        // The rare block then redoes the reference's tests in the reference's own terms and order.
        // (a wave none of whose rays is inside the radius - the last dozen attempts of every outgoing ray - skips all of it and tests
        // the velocity, as the fixed-step loop does)
        float q, nds1;
        unsigned long long rare_lanes;   // (the ballot taken where the compares are: a lane mask that crosses the branch as a bool comes back through a register)
        if (wave_inside) {
            q = controller.raw_error_q(next_acceleration);
            nds1 = __builtin_amdgcn_fmed3f(controller.root99 * __builtin_amdgcn_rsqf(__builtin_sqrtf(q)), (0.99f * 0.3f) * near_ds, (0.99f * 2.f) * near_ds);
            const float clamped = __builtin_amdgcn_fmed3f(nds1, controller.floor_step, controller.ambient);
            const float threshold = near_ds * (1 / 1.95f);
            // (one ballot per compare, joined as scalars: the ballot of an OR of compares is lowered through a register)
            rare_lanes = __builtin_amdgcn_ballot_w64(!(__builtin_fmaf(q, 0.f, clamped) >= threshold));
#ifdef SINGULARITY_DETECTION
            rare_lanes |= __builtin_amdgcn_ballot_w64(clamped == controller.floor_step);   // (with min_step > ambient - nobody's setting - every attempt takes the rare block: slow, exact)
#endif
            if (!step_controller::error_sees_every_component) {   // a metric with a zero weight: the velocity's own test as well
                const float poison = (next_velocity.x + next_velocity.y) + (next_velocity.z + next_velocity.w);
                rare_lanes |= __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(poison) <= 3.402823466e+38f));
            }
            next_ds = inside ? clamped : near_ds;   // also on an attempt that is then rejected (cl.cl:3443 comes before the returns)
        } else {
            // no ray of the wave inside the radius: nothing of the controller applies; a non-finite velocity (their sum) is handed to
            // the rare block as a NaN "error"
            const float poison = (next_velocity.x + next_velocity.y) + (next_velocity.z + next_velocity.w);
            q = poison * 0.f;
            nds1 = 0;
            rare_lanes = __builtin_amdgcn_ballot_w64(!(q == q));
        }
        if (__builtin_expect(rare_lanes != 0, 0)) {
            asm volatile("; rejected step, singularity, degenerate state");
            const float nds = max_f32_uniform(controller.min_step, nds1);
#ifdef SINGULARITY_DETECTION
            dead = inside & (q > controller.singular_q) & (nds == controller.min_step);   // DS_RETURN: lost
#endif
            bool reject = inside & (nds < ds * (1 / 1.95f));   // DS_SKIP: retry from the same state with the smaller step
            // IS_DEGENERATE on the accepted state.  A rejected attempt is not tested (the reference `continue`s before its test: an
            // overshoot into a singularity is retried with the smaller step).
            bool degenerate = !(q == q);
            if (!step_controller::error_sees_every_component) {
                const float poison = (next_velocity.x + next_velocity.y) + (next_velocity.z + next_velocity.w);
                degenerate |= !(__builtin_fabsf(poison) <= 3.402823466e+38f);
            }
            dead |= !reject & degenerate;
#ifndef GR_ACCEL_WITHOUT_TRIG   // (capi.cpp accelerations_without_trig: a program whose accelerations call no sin / cos has no such NaN to meet - and a
                                // Kerr-Schild ray that overshoots the ring is retried right here, not sent to the slow loop for the rest of its 16 384 steps)
            if (!decltype(libm)::value) {
                // The polynomial sin / cos poisons an argument outside its range with a NaN (metric.hip) - which the reference, with a
                // full-range sine, would never have seen.  Such an attempt must not run through the controller (v_med3 hands a NaN error
                // back as the smallest step: a rejection, then ~10 more down to min_step, `running` multiplied by a garbage K on each
                // with reparameterisation on): a NaN on the fast path leaves the loop AT ONCE, nothing rejected, and the attempt is done
                // again below with libm from the step and the `running` it started with - where a NaN that is the metric's own (a
                // singularity overshot) then takes the reference's course.  
                dead |= degenerate;
                reject &= !degenerate;
            }
#endif
            if (reject) {
                overwrite(po, position);
                overwrite(vo, velocity);
                overwrite(ao, acceleration);
                asm volatile("v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1" : "+v"(rejections), "+v"(budget));   // the step it did not take
            }
        }
#else
        {
            // IS_DEGENERATE on the accepted velocity: a non-finite sum <=> a non-finite component (finite components cannot
            // overflow the sum below ~1e38); a non-finite acceleration makes the velocity computed from it non-finite in the same step
            const float poison = (next_velocity.x + next_velocity.y) + (next_velocity.z + next_velocity.w);
            dead = !(__builtin_fabsf(poison) <= 3.402823466e+38f);
        }
#endif
        // the rare exit: the state the ray is left in, (p, v, a), has just passed the loop-top tests
        if (__builtin_expect(dead, 0)) return true;
        if (RESUMABLE) {
            // lanes still in the loop = the exec mask; fewer than keep_lanes of them: leave and let the caller refill the wave
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < keep_lanes) { pause_wave = true; return true; }
        }
        return false;
    };

    // The state a ray leaves the fast loop in stays in register set 0 (below): no registers of its own.  (Earlier forms: left to
    // the compiler, "whichever set the ray was in when it left" becomes twelve running copies per attempt; in twelve registers of
    // its own it cost a wave per SIMD or, held to six waves, 60 bytes of scratch per lane and 0.1-0.2 GB of scratch traffic per 4K
    // launch; parked in LDS by hand - 14 KB per workgroup - it measured 6 % slower than the spill.)
    float exit_ds = 0, exit_running = 1;
    {
        const trig_flavour<false> polynomial;
#if GR_PRIORITY_TRIPS > 0
        unsigned trips = 0, next_level = GR_PRIORITY_TRIPS;
        // (the wave of rays that were parked: GR_PARK_PRIORITY for the visit.  0, on measurement - DESIGN.md section 4)
#ifndef GR_PARK_PRIORITY
#define GR_PARK_PRIORITY 0
#endif
        if (PARKABLE && resumed) { __builtin_amdgcn_s_setprio(GR_PARK_PRIORITY); next_level = 0xffffffffu; }
#endif
        // (wave-uniform, and said to be: the test below is then three scalar instructions)
        unsigned int visit_trips = 0;
        const int park_below = PARKABLE ? __builtin_amdgcn_readfirstlane(keep_lanes) : 0;
        const unsigned int park_after = PARKABLE ? __builtin_amdgcn_readfirstlane(park_trips) : 0u;
        for (;;) {
            float4 p1, v1, a1;
            float ds_used, running_before;
#if GR_PRIORITY_TRIPS > 0
            // A wave whose rays are still going after GR_PRIORITY_TRIPS trips (two attempts each) is on the launch's critical path -
            // a launch lasts at least as long as its longest ray, and a ray's attempts are a dependent chain that gets one issue
            // slot in (waves per SIMD) while the SIMD is full.  The hardware's issue arbiter takes the highest-priority ready wave
            // first: raising the priority of the long waves (again at twice and four times the count) shortens their chain towards
            // what a wave alone on its SIMD achieves and costs the others exactly the slots it takes - the same work, a shorter tail.
            // Wave-uniform: the trip count is the same for every lane still in the loop.  Arithmetic untouched.
            if (__builtin_expect(++trips == next_level, 0)) {
                if (next_level == GR_PRIORITY_TRIPS) __builtin_amdgcn_s_setprio(1);
                else if (next_level == 2 * GR_PRIORITY_TRIPS) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
                next_level = next_level < 4 * GR_PRIORITY_TRIPS ? 2 * next_level : 0xffffffffu;
            }
#endif
            // A ray that leaves is left in set 0: lanes that have left are masked off for the rest of the loop, so set 0 keeps their
            // state with no registers of its own; a ray that leaves from set 1 is copied over first (opaque copies, once per ray).
            if (attempt(polynomial, p0, v0, a0, p1, v1, a1, ds_used, running_before)) {
                if (RESUMABLE && pause_wave) { overwrite(p0, p1); overwrite(v0, v1); overwrite(a0, a1); }
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "+v"(exit_ds), "+v"(exit_running) : "v"(ds_used), "v"(running_before));
                break;
            }
            if (attempt(polynomial, p1, v1, a1, p0, v0, a0, ds_used, running_before)) {
                if (!(RESUMABLE && pause_wave)) { overwrite(p0, p1); overwrite(v0, v1); overwrite(a0, a1); }
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "+v"(exit_ds), "+v"(exit_running) : "v"(ds_used), "v"(running_before));
                break;
            }
            if (PARKABLE) {
                // the trip is complete, the state in set 0: a wave down to its last few rays after a long time hands them over (scalar
                // work only: a count of the exec mask and two compares)
                visit_trips = __builtin_amdgcn_readfirstlane(visit_trips) + 1u;
                if (__builtin_expect(visit_trips >= park_after && __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < park_below, 0)) {
                    pause_wave = true;
                    break;
                }
            }
        }
    }
#if GR_PRIORITY_TRIPS > 0
    __builtin_amdgcn_s_setprio(0);
#endif
    float4 position = p0, velocity = v0, acceleration = a0;
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    // Why the loop was left is read off what it leaves behind (flags set inside it would have to be carried through it per lane):
    // an exhausted step budget has wrapped around; otherwise the loop-top tests on the final state, in the reference's order
    // (cl.cl:3990-4130) - every other exit leaves a state that has just passed them.
    bool capped, lost_at_top, terminated_at_top;
    auto classify = [&]() {
        capped = budget == 0xffffffffu;
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        lost_at_top = stop_lost(position, velocity, acceleration, running);
        terminated_at_top = stop_terminated(position, polar);
    };
    classify();
#if !defined(GR_FAST_TRIG) && !defined(GR_LIBM_TRIG) && !defined(GR_PROBE_NO_SLOW_TRIG)
    if (__builtin_expect(!capped && !pause_wave && !(lost_at_top | terminated_at_top), 0)) {
        // Left at the bottom of an attempt: degenerate for good, or a sin / cos argument outside the polynomial's range.  The
        // attempt is done again, and the ray integrated on if it was the latter, one attempt per trip with sin / cos from libm.
        // What the loop carries is put back to what it was before the abandoned attempt.
        const float ds_used = exit_ds;
        float4 polar = gm::generic_to_spherical(position, cfg);
        if (__builtin_fabsf(gm::distance_to_object_from(position, polar, cfg)) < new_max) next_ds = ds_used;   // min(next_ds, ambient) gives ds_used again
        running = exit_running;
        budget++;
        const trig_flavour<true> precise;
        for (;;) {
            float4 np, nv, na;
            float unused_ds, unused_running;
            const bool leave = attempt(precise, position, velocity, acceleration, np, nv, na, unused_ds, unused_running);
            if (leave && !pause_wave) break;
            position = np; velocity = nv; acceleration = na;
            if (leave) break;
        }
        classify();
    }
#endif
    paused = (RESUMABLE || PARKABLE) && pause_wave;
    const bool left_at_top = !capped && !paused && (lost_at_top | terminated_at_top);
    int result = RAY_LOST;
    if (!paused) {
        // a state that is degenerate anywhere is the reference's plain `return` (cl.cl:4235-4244, terminated stays 0) even when its
        // position happens to lie beyond the boundary
        const bool finite = degenerate_accumulate(position, degenerate_accumulate(velocity, degenerate_accumulate(acceleration, 0.f))) == 0.f;
        if (!capped && !lost_at_top && terminated_at_top && finite) result = RAY_TERMINATED;
    }
    // every attempt() entered took one step off the budget; the entry that found the ray finished (or the budget empty) made none
    const unsigned int taken = capped ? budget_before : budget_before - budget - (left_at_top ? 1u : 0u);
    if (RESUMABLE || PARKABLE) { s.next_ds = next_ds; s.steps += (int)taken; s.tries += taken + rejections; }
    s.position = position;
    s.velocity = velocity;
    s.acceleration = acceleration;
    s.running_dlambda_dnew = running;
    if (attempts) *attempts = (RESUMABLE || PARKABLE) ? s.tries : taken + rejections;
    return result;
}

__device__ __forceinline__ int integrate_ray(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts) {
    bool paused;
    return integrate_pingpong<false>(s, cfg, dfg, attempts, 0, paused);
}


#ifdef GR_TWO_RAYS_PER_LANE
// ---- two rays per lane ---------------------------------------------------------------------------
// integrate_pingpong's algorithm with every per-ray float held as a pair (ray 0 in the low, ray 1 in the high half of a 64-bit register
// pair).  Measured on MI355X (tools/ubench/accel_rate.hip: the substituted Kerr acceleration + Verlet update alone): packed
// instructions get through two rays' arithmetic in less issue time than two plain ones, 374 -> 462 G ray-steps/s.  Three
// quarters of the instructions of a Verlet attempt are such multiplies and fmas, so one lane stepping two rays gets through more
// attempts per cycle.  What has no packed form (compares, selects, rcp/rsq/sqrt, the commit of an accepted step) is done per
// half.  The arithmetic of a ray is that of the one-ray loop (integrate_pingpong); a ray that has left the loop keeps
// its state (the commit is per ray) while its partner goes on.
using gm::pairf;
using gm::pair4;
using gm::splat;
__device__ __forceinline__ pair4 operator+(pair4 a, pair4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ pair4 operator*(pair4 a, pairf s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
template <int H> __device__ __forceinline__ float4 half_of(pair4 v) {
    return H == 0 ? f4(v.x.x, v.y.x, v.z.x, v.w.x) : f4(v.x.y, v.y.y, v.z.y, v.w.y);
}
__device__ __forceinline__ pair4 pair_of(float4 a, float4 b) {
    pair4 r;
    r.x.x = a.x; r.x.y = b.x; r.y.x = a.y; r.y.y = b.y; r.z.x = a.z; r.z.y = b.z; r.w.x = a.w; r.w.y = b.w;
    return r;
}

#ifdef ADAPTIVE_PRECISION
__device__ __forceinline__ pairf acceleration_to_precision(pair4 acc, float max_acceleration, pairf& next_ds) {
    pair4 wa = {acc.x * (float)(W_V1), acc.y * (float)(W_V2), acc.z * (float)(W_V3), acc.w * (float)(W_V4)};
    pairf d2 = wa.x * wa.x + wa.y * wa.y + wa.z * wa.z + wa.w * wa.w;
    pairf current;
    current.x = __builtin_sqrtf(d2.x); current.y = __builtin_sqrtf(d2.y);
    current = current * 0.01f;
    current = current / GR_W_MAX;
    const float scale = 65536.f;
    float err = max_acceleration;
    pairf diff = current * scale;
    float floor_diff = err * scale / 1e10f;
    if (diff.x < floor_diff) diff.x = floor_diff;
    if (diff.y < floor_diff) diff.y = floor_diff;
    const float root = __builtin_sqrtf(err * scale);
    next_ds.x = root * __builtin_amdgcn_rsqf(diff.x);
    next_ds.y = root * __builtin_amdgcn_rsqf(diff.y);
    return diff;
}
#endif

// in: the initial states of the two rays (an inactive ray carries a copy of its partner's so that its half computes on
// benign numbers); out: final position, velocity, running_dlambda_dnew, outcome and attempts per ray
__device__ __forceinline__ void integrate_pair(pair4& position_io, pair4& velocity_io, pair4 acceleration, pairf& running_out,
                                               bool active0, bool active1, cfg_t cfg, dfg_t dfg, int& result0, int& result1,
                                               unsigned int& tries0, unsigned int& tries1) {
    pair4 position = position_io, velocity = velocity_io;
    pairf f_in_x;
    f_in_x.x = __builtin_fabsf(velocity.x.x); f_in_x.y = __builtin_fabsf(velocity.x.y);
#ifdef IS_CONSTANT_THETA
    position.z = splat(GR_PIf / 2); velocity.z = splat(0.f); acceleration.z = splat(0.f);
#endif
    pairf next_ds = splat(0.00001f);
#ifdef ADAPTIVE_PRECISION
    const float max_accel = GET_FEATURE(max_acceleration_change, dfg);
    const float min_step = GET_FEATURE(min_step, dfg);
    (void)acceleration_to_precision(acceleration, max_accel, next_ds);
#endif
    const float subambient_precision = 0.5f;
    const float ambient_precision = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg);
    const float new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    pairf running = splat(1.f);
    const int loop_limit = 4096 * 4;
    unsigned int t0 = 0, t1 = 0;
    int i0 = 0, i1 = 0;
    bool alive0 = active0, alive1 = active1;

    // the loop-top tests of the one-ray loop on one ray's numbers
    auto stop_lost = [&](float pos_y, float vel_x_over_run, float acc_x_over_run, float fin, int steps) {
        bool lost = steps >= loop_limit;
#ifdef HAS_CYLINDRICAL_SINGULARITY
        lost |= pos_y < CYLINDRICAL_TERMINATOR;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        lost |= __builtin_fabsf(vel_x_over_run) > 1000 + fin && __builtin_fabsf(acc_x_over_run) > 100;
#endif
        (void)pos_y; (void)vel_x_over_run; (void)acc_x_over_run; (void)fin;
        return lost;
    };
    auto stop_terminated = [&](float polar_y) {
        bool t = __builtin_fabsf(polar_y) >= universe;
#ifdef SINGULAR
        t |= __builtin_fabsf(polar_y) < SINGULAR_TERMINATOR;
#endif
        return t;
    };
    for (;;) {
#ifdef IS_CONSTANT_THETA
        position.z = splat(GR_PIf / 2); velocity.z = splat(0.f); acceleration.z = splat(0.f);
#endif
        pair4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = splat(GR_PIf / 2);
#endif
        pairf r_value = gm::distance_to_object_from(position, polar, cfg);
        pairf ar;
        ar.x = __builtin_fabsf(r_value.x); ar.y = __builtin_fabsf(r_value.y);
        pairf ds;
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#else
        ds.x = mixf(ambient_precision, subambient_precision, (clampf(ar.x, new_min, new_max) - new_min) / (new_max - new_min));
        ds.y = mixf(ambient_precision, subambient_precision, (clampf(ar.y, new_min, new_max) - new_min) / (new_max - new_min));
#endif
        if (ar.x < new_max) ds.x = __builtin_fminf(ds.x, ambient_precision);
        else ds.x = 0.1f * (ar.x - new_max) + ambient_precision;
        if (ar.y < new_max) ds.y = __builtin_fminf(ds.y, ambient_precision);
        else ds.y = 0.1f * (ar.y - new_max) + ambient_precision;

#ifndef UNCONDITIONALLY_NONSINGULAR
        const pairf vq = velocity.x / running, aq = acceleration.x / running;
#else
        const pairf vq = splat(0.f), aq = splat(0.f);
#endif
        alive0 = alive0 && !(stop_lost(position.y.x, vq.x, aq.x, f_in_x.x, i0) | stop_terminated(polar.y.x));
        alive1 = alive1 && !(stop_lost(position.y.y, vq.y, aq.y, f_in_x.y, i1) | stop_terminated(polar.y.y));
        if (!(alive0 | alive1)) break;
        t0 += alive0 ? 1u : 0u;
        t1 += alive1 ? 1u : 0u;

        // velocity Verlet (step_verlet), both rays
        const pairf half_ds = ds * 0.5f, half_ds2 = half_ds * ds;
        pair4 next_position = position + velocity * ds + acceleration * half_ds2;
        pair4 half_velocity = velocity + acceleration * ds;
        pair4 next_acceleration = GR_PROBE_ACCELERATION_PAIR(gm::geodesic_acceleration(next_position, half_velocity, cfg), acceleration, half_velocity, next_position);
        pair4 next_velocity = velocity + (acceleration + next_acceleration) * half_ds;
        if (reparam) {
            pairf K;
            K.x = 1 / __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x.x), __builtin_fabsf(next_velocity.y.x)),
                                      __builtin_fmaxf(__builtin_fabsf(next_velocity.z.x), __builtin_fabsf(next_velocity.w.x)));
            K.y = 1 / __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x.y), __builtin_fabsf(next_velocity.y.y)),
                                      __builtin_fmaxf(__builtin_fabsf(next_velocity.z.y), __builtin_fabsf(next_velocity.w.y)));
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
            if (alive0) running.x *= K.x;
            if (alive1) running.y *= K.y;
        }

        bool accept0 = true, accept1 = true;
#ifdef ADAPTIVE_PRECISION
        {
            // calculate_ds_error for both rays in one straight line: every step below is two independent instructions (or
            // one packed one), so the serial chain sqrt -> rsq -> clamp -> compare at the end of an attempt is walked once for
            // the two rays, not once per ray, and nothing in it changes the exec mask
            pairf suggested;
            pairf diff = acceleration_to_precision(next_acceleration, max_accel, suggested);
            const pairf want = suggested * 0.99f, lo = ds * (0.99f * 0.3f), hi = ds * (0.99f * 2.f), back = ds / 1.95f;
            pairf nds;
            nds.x = __builtin_fmaxf(clampf(want.x, lo.x, hi.x), min_step);
            nds.y = __builtin_fmaxf(clampf(want.y, lo.y, hi.y), min_step);
            const bool inside0 = ar.x < new_max, inside1 = ar.y < new_max;
            next_ds.x = inside0 ? nds.x : next_ds.x;
            next_ds.y = inside1 ? nds.y : next_ds.y;
#ifdef SINGULARITY_DETECTION
            const pairf dq = diff / 65536.f;
            alive0 = alive0 & !(inside0 & (nds.x == min_step) & (dq.x > max_accel * 10000));
            alive1 = alive1 & !(inside1 & (nds.y == min_step) & (dq.y > max_accel * 10000));
#endif
            accept0 = !inside0 | !(nds.x < back.x);   // back-step: retry from the same state with the smaller step
            accept1 = !inside1 | !(nds.y < back.y);
            (void)diff;
        }
#endif
        if (alive0 && accept0) {
            position.x.x = next_position.x.x; position.y.x = next_position.y.x; position.z.x = next_position.z.x; position.w.x = next_position.w.x;
            velocity.x.x = next_velocity.x.x; velocity.y.x = next_velocity.y.x; velocity.z.x = next_velocity.z.x; velocity.w.x = next_velocity.w.x;
            acceleration.x.x = next_acceleration.x.x; acceleration.y.x = next_acceleration.y.x; acceleration.z.x = next_acceleration.z.x; acceleration.w.x = next_acceleration.w.x;
            i0++;
            float poison = degenerate_accumulate(half_of<0>(position), degenerate_accumulate(half_of<0>(velocity), 0.f));
            if (reparam) poison = degenerate_accumulate(half_of<0>(acceleration), poison);
            if (!(poison == 0.f)) alive0 = false;
        }
        if (alive1 && accept1) {
            position.x.y = next_position.x.y; position.y.y = next_position.y.y; position.z.y = next_position.z.y; position.w.y = next_position.w.y;
            velocity.x.y = next_velocity.x.y; velocity.y.y = next_velocity.y.y; velocity.z.y = next_velocity.z.y; velocity.w.y = next_velocity.w.y;
            acceleration.x.y = next_acceleration.x.y; acceleration.y.y = next_acceleration.y.y; acceleration.z.y = next_acceleration.z.y; acceleration.w.y = next_acceleration.w.y;
            i1++;
            float poison = degenerate_accumulate(half_of<1>(position), degenerate_accumulate(half_of<1>(velocity), 0.f));
            if (reparam) poison = degenerate_accumulate(half_of<1>(acceleration), poison);
            if (!(poison == 0.f)) alive1 = false;
        }
    }
    // why each ray left the loop: the loop-top tests on its final state (as integrate_pingpong's classify)
    {
        pair4 polar = gm::generic_to_spherical(position, cfg);
#ifndef UNCONDITIONALLY_NONSINGULAR
        const pairf vq = velocity.x / running, aq = acceleration.x / running;
#else
        const pairf vq = splat(0.f), aq = splat(0.f);
#endif
        const bool finite0 = degenerate_accumulate(half_of<0>(position), degenerate_accumulate(half_of<0>(velocity), degenerate_accumulate(half_of<0>(acceleration), 0.f))) == 0.f;
        const bool finite1 = degenerate_accumulate(half_of<1>(position), degenerate_accumulate(half_of<1>(velocity), degenerate_accumulate(half_of<1>(acceleration), 0.f))) == 0.f;
        result0 = (!stop_lost(position.y.x, vq.x, aq.x, f_in_x.x, i0) && stop_terminated(polar.y.x) && finite0) ? RAY_TERMINATED : RAY_LOST;
        result1 = (!stop_lost(position.y.y, vq.y, aq.y, f_in_x.y, i1) && stop_terminated(polar.y.y) && finite1) ? RAY_TERMINATED : RAY_LOST;
    }
    position_io = position;
    velocity_io = velocity;
    running_out = running;
    tries0 = t0;
    tries1 = t1;
}
#endif  // GR_TWO_RAYS_PER_LANE

