// quadsim_device.hpp -- per-env device math of the fused docking step (gfx950).
//
// One lane integrates one env: chaser + target drone, target PID, dock ports,
// relative observation, reward, done.  All state lives in VGPRs between the
// tile load and the tile store.  Formulas follow SURVEY.md Appendix A, which is
// the validated closed form of
//   dynamics/quadrotor.py, utils/transform.py, controller/PIDController.py,
//   gym-docking/gym_docking/envs/{docking_env,moving_docking_env}.py
// (file:line cited per function).  fp32 throughout; where binary32 would lose
// digits to cancellation the evaluation order is chosen for accuracy, never to
// mimic the reference's float64 rounding (parity is to 1e-5, not bitwise).
#pragma once

#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <stdint.h>

namespace qs {

constexpr float kG = 9.81f;                       // dynamics/quadrotor.py:15
constexpr float kL = 0.086f;                      // :20
constexpr float kLambda = (float)(1.5e-9 / 6.11e-8);  // km/kf, :43-45
constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kLim85 = 1.4835298641951802f;     // deg2rad(85),  quadrotor.py:156
constexpr float kLim175 = 3.0543261909900763f;    // deg2rad(175), quadrotor.py:162
constexpr float kLim10 = 0.17453292519943295f;    // deg2rad(10),  docking_env.py:132
constexpr float kTMax = 600.0f;                   // docking_env.py:152

// AoSoA tile: 64 envs (one wavefront) x kRecWords fields, field-major inside the tile
constexpr int kTile = 64;
constexpr int kRecWords = 40;
constexpr int kParWords = 4;
// field offsets inside an env record (same order as the oracle's "rec")
constexpr int F_SC = 0, F_ST = 13, F_UC = 26, F_UT = 30, F_QD = 34, F_LS = 38, F_T = 39;

enum : unsigned { FLAG_DOCKED = 1, FLAG_OVERLIMIT = 2, FLAG_OVERTIME = 4, FLAG_CLIM = 8, FLAG_TLIM = 16 };
enum : uint64_t { STREAM_AUTORESET = 0, STREAM_RESET = 1, STREAM_ACTIONS = 2, STREAM_CTOR = 3, STREAM_POLICY = 4 };

struct Par {
    float m, Ixx, Iyy, Izz;
};

// env-kind constants, uniform over the launch (SGPRs)
struct EnvConst {
    int kind;      // target controller: 0 controller.PID (docking-v0/v1), 1 vel_controller (docking-v2)
    float dt;
    float rmax;    // 3 (docking_env.py:141) / 10 (moving_docking_env.py:148)
    float vdes_x;  // 0 / 0.2 (moving_docking_env.py:62)
};

// ---------------------------------------------------------------------------
// scalar math.  Hardware rcp / rsq / sqrt (1 ulp) and hand-rolled inverse trig /
// sincos: the arguments on this path are bounded (angles in [-pi, pi], ratios in
// [0, 1]), so none of libm's huge-argument or denormal paths are needed.
// Polynomials are near-minimax fits (tools/fit_polys.py); measured against
// float64 on the full argument range: atan2 <= 1.1e-7 abs, asin <= 9.3e-8 abs,
// sin <= 4.4e-8, cos <= 7.4e-8 abs -- about 1 ulp of the result's scale.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float q_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float q_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float q_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// asin on [-1, 1]: x + x z P(z) for |x| <= 1/2, pi/2 - 2 asin(sqrt((1-|x|)/2)) above
__device__ __forceinline__ float q_asin(float x)
{
    float a = fabsf(x);
    bool big = a > 0.5f;
    float z = big ? fmaf(-0.5f, a, 0.5f) : a * a;
    float s = big ? q_sqrt(z) : a;
    float p = 0.038206227123737335f;
    p = fmaf(p, z, 0.026494402438402176f);
    p = fmaf(p, z, 0.04501067474484444f);
    p = fmaf(p, z, 0.07498809695243835f);
    p = fmaf(p, z, 0.16666673123836517f);
    float r = fmaf(s * z, p, s);
    r = big ? fmaf(-2.0f, r, kHalfPi) : r;
    return copysignf(r, x);
}

// atan2 for finite arguments; atan2(0, 0) = 0 like numpy.  atan(t) = t + t s P(s), s = t^2, t in [0, 1].
__device__ __forceinline__ float q_atan2(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = mn * q_rcp(mx);
    t = (mx == 0.0f) ? 0.0f : t;
    float s = t * t;
    float p = 0.0028340641874819994f;
    p = fmaf(p, s, -0.016005029901862144f);
    p = fmaf(p, s, 0.042587608098983765f);
    p = fmaf(p, s, -0.07495445758104324f);
    p = fmaf(p, s, 0.10636754333972931f);
    p = fmaf(p, s, -0.14202570915222168f);
    p = fmaf(p, s, 0.19992484152317047f);
    p = fmaf(p, s, -0.3333306610584259f);
    float r = fmaf(t * s, p, t);
    r = (ay > ax) ? (kHalfPi - r) : r;
    r = (x < 0.0f) ? (kPi - r) : r;
    return copysignf(r, y);
}

// sin and cos for |x| <~ 1e4: two-constant Cody-Waite reduction by pi/2 (exact with fma),
// degree-7 / degree-8 polynomials on [-pi/4, pi/4], quadrant fix-up by sign-bit arithmetic.
__device__ __forceinline__ void q_sincos(float x, float &sn, float &cs)
{
    float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(kf, -1.57079637050628662109375f, x);
    r = fmaf(kf, 4.37113900018624283e-8f, r);
    int k = (int)kf;
    float z = r * r;
    float sp = fmaf(fmaf(-0.0001958291686605662f, z, 0.008332724682986736f), z, -0.166666641831398f);
    float s0 = fmaf(r * z, sp, r);
    float cp = fmaf(fmaf(2.4542947357986122e-05f, z, -0.0013888279208913445f), z, 0.0416666641831398f);
    float c0 = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
    bool swap = (k & 1) != 0;
    float ss = swap ? c0 : s0, cc = swap ? s0 : c0;
    sn = __uint_as_float(__float_as_uint(ss) ^ (((unsigned)k & 2u) << 30));
    cs = __uint_as_float(__float_as_uint(cc) ^ (((unsigned)(k + 1) & 2u) << 30));
}

// sin and cos for |x| <= pi/4 (no reduction, no quadrant logic): the same polynomials as q_sincos
__device__ __forceinline__ void q_sincos_small(float r, float &sn, float &cs)
{
    float z = r * r;
    float sp = fmaf(fmaf(-0.0001958291686605662f, z, 0.008332724682986736f), z, -0.166666641831398f);
    sn = fmaf(r * z, sp, r);
    float cp = fmaf(fmaf(2.4542947357986122e-05f, z, -0.0013888279208913445f), z, 0.0416666641831398f);
    cs = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
}

// off-diagonal entries of the reference's quat2rot (diagonal is identically 1):
// utils/transform.py:4-20 == dynamics/quadrotor.py:226-245.  Element-wise
// qa_hat*qa_hat with the NORMALISED vector part, linear term with the
// UN-normalised scalar part.
struct Rot {
    float r01, r02, r10, r12, r20, r21;
};
__device__ __forceinline__ Rot quat2rot(const float q[4])
{
    float inv = q_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float n1 = q[1] * inv, n2 = q[2] * inv, n3 = q[3] * inv;
    float w2 = 2.0f * q[0];
    float s1 = 2.0f * n1 * n1, s2 = 2.0f * n2 * n2, s3 = 2.0f * n3 * n3;
    Rot R;
    R.r01 = s3 - w2 * n3;
    R.r10 = s3 + w2 * n3;
    R.r02 = s2 + w2 * n2;
    R.r20 = s2 - w2 * n2;
    R.r12 = s1 - w2 * n1;
    R.r21 = s1 + w2 * n1;
    return R;
}

// utils/transform.py:94-120.  The three branches collapse to: phi = asin(clamp r12),
// theta = 0 when r12 >= 1 or r12 < -1.
__device__ __forceinline__ void quat2euler(const float q[4], float &phi, float &theta, float &psi)
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float r10 = 2.0f * (x * y - w * z);
    float r11 = w * w - x * x + y * y - z * z;
    float r12 = 2.0f * (w * x + y * z);
    float r02 = 2.0f * (x * z - w * y);
    float r22 = w * w - x * x - y * y + z * z;
    psi = q_atan2(-r10, r11);
    phi = q_asin(fminf(fmaxf(r12, -1.0f), 1.0f));
    float th = q_atan2(-r02, r22);
    theta = (r12 >= 1.0f || r12 < -1.0f) ? 0.0f : th;
}

// sin/cos of the yaw psi = atan2(-r10, r11) of quat2euler and of psi/2, by algebra instead of
// atan2 + sincos: (cos psi, sin psi) = (r11, -r10)/hypot, half angles from the branch that
// does not cancel.  (What hover/vel_controller read back from state_des, PIDController.py:87-91,100.)
__device__ __forceinline__ void quat_yaw_trig(const float q[4], float &sp, float &cp, float &sh, float &ch)
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float r10 = 2.0f * (x * y - w * z);
    float r11 = w * w - x * x + y * y - z * z;
    float h2 = r10 * r10 + r11 * r11;
    float ih = q_rsqrt(h2);
    bool degenerate = !(h2 > 0.0f);                       // atan2(0,0) = 0
    sp = degenerate ? 0.0f : -r10 * ih;
    cp = degenerate ? 1.0f : r11 * ih;
    // half angle from the branch that does not cancel: cos(psi/2) when cos psi >= 0, |sin(psi/2)| otherwise
    bool pos = cp >= 0.0f;
    float big = q_sqrt(fmaf(pos ? 0.5f : -0.5f, cp, 0.5f));
    float small = 0.5f * fabsf(sp) * q_rcp(big);
    ch = pos ? big : small;
    sh = copysignf(pos ? small : big, sp);
}

// euler2quat from precomputed half-angle sines / cosines (utils/transform.py:123-136)
__device__ __forceinline__ void euler2quat_trig(float sr, float cr, float sp, float cp, float sy, float cy, float q[4])
{
    q[0] = cr * cp * cy - sr * sp * sy;
    q[1] = sr * cp * cy - cr * sp * sy;
    q[2] = sr * cp * sy + cr * sp * cy;
    q[3] = cr * cp * sy + sr * sp * cy;
}

// utils/transform.py:123-136
__device__ __forceinline__ void euler2quat(float roll, float pitch, float yaw, float q[4])
{
    float sy, cy, sp, cp, sr, cr;
    q_sincos(yaw * 0.5f, sy, cy);
    q_sincos(pitch * 0.5f, sp, cp);
    q_sincos(roll * 0.5f, sr, cr);
    q[0] = cr * cp * cy - sr * sp * sy;
    q[1] = sr * cp * cy - cr * sp * sy;
    q[2] = sr * cp * sy + cr * sp * cy;
    q[3] = cr * cp * sy + sr * sp * cy;
}

// Drone.df, dynamics/quadrotor.py:80-113 (R[2,2] == 1, diagonal inertia)
__device__ __forceinline__ void drone_df(const float s[13], const float u[4], const Par &P, float inv_m, float ds[13])
{
    const float *q = s + 6, *w = s + 10;
    Rot R = quat2rot(q);
    float Fm = u[0] * inv_m;
    ds[0] = s[3]; ds[1] = s[4]; ds[2] = s[5];
    ds[3] = R.r20 * Fm;
    ds[4] = R.r21 * Fm;
    ds[5] = Fm - kG;
    float eq = 1.0f - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float kq = 2.0f * eq;                                   // K_quat * e_quat, :96-97
    float k0 = -w[0] * q[1] - w[1] * q[2] - w[2] * q[3];    // :98
    float k1 = w[0] * q[0] - w[1] * q[2] + w[2] * q[3];     // :99
    float k2 = w[1] * q[0] + w[2] * q[1] - w[0] * q[3];     // :100
    float k3 = w[2] * q[0] - w[1] * q[1] + w[0] * q[2];     // :101
    ds[6] = -0.5f * k0 + kq * q[0];
    ds[7] = -0.5f * k1 + kq * q[1];
    ds[8] = -0.5f * k2 + kq * q[2];
    ds[9] = -0.5f * k3 + kq * q[3];
    // inv(I) (I u[1:] - w x I w), :83,:105
    float Iw0 = P.Ixx * w[0], Iw1 = P.Iyy * w[1], Iw2 = P.Izz * w[2];
    float c0 = w[1] * Iw2 - w[2] * Iw1;
    float c1 = w[2] * Iw0 - w[0] * Iw2;
    float c2 = w[0] * Iw1 - w[1] * Iw0;
    ds[10] = u[1] - c0 * q_rcp(P.Ixx);
    ds[11] = u[2] - c1 * q_rcp(P.Iyy);
    ds[12] = u[3] - c2 * q_rcp(P.Izz);
}

// Drone.attitude_limit + write-back, dynamics/quadrotor.py:146-168,:135-138.
// The three threshold tests are made on the matrix entries instead of the angles
// (asin and atan2 are monotone): |roll| >= 85deg <=> |clamp r12| >= sin 85deg;
// |pitch| >= 85deg <=> r22 <= |r02| cot 85deg (pitch is 0 on the saturated branches);
// |yaw| >= 175deg <=> r11 < 0 and |r10| <= -r11 tan 5deg.  The angles themselves are only
// evaluated in the rare branch where a limit fires.  Sequential overriding ifs of the
// reference == the LAST violated axis wins, built from the un-clamped other two angles.
__device__ __forceinline__ bool attitude_limit(float s[13])
{
    constexpr float kSin85 = 0.99619469809174555f;
    constexpr float kTan5 = 0.087488663525924005f;
    float w = s[6], x = s[7], y = s[8], z = s[9];
    float r10 = 2.0f * (x * y - w * z);
    float r11 = w * w - x * x + y * y - z * z;
    float r12 = 2.0f * (w * x + y * z);
    float r02 = 2.0f * (x * z - w * y);
    float r22 = w * w - x * x - y * y + z * z;
    bool sat = (r12 >= 1.0f) || (r12 < -1.0f);
    bool a = fabsf(r12) >= kSin85;
    bool b = !sat && (r22 <= fabsf(r02) * kTan5) && !(r02 == 0.0f && r22 == 0.0f);
    bool c = (r11 < 0.0f) && (fabsf(r10) <= -r11 * kTan5);
    bool over = a || b || c;
    if (over) {
        float r = q_asin(fminf(fmaxf(r12, -1.0f), 1.0f));
        float p = sat ? 0.0f : q_atan2(-r02, r22);
        float yw = q_atan2(-r10, r11);
        if (c) yw = copysignf(kLim175, yw);
        else if (b) p = copysignf(kLim85, p);
        else r = copysignf(kLim85, r);
        euler2quat(r, p, yw, s + 6);
        s[10] = 0.0f; s[11] = 0.0f; s[12] = 0.0f;
    }
    return over;
}

// Drone.u_limit, dynamics/quadrotor.py:171-187 (A :47-50, B :52-54); per-rotor clamp [0, m g]
__device__ __forceinline__ void u_limit(const float u[4], float mg, float out[4])
{
    constexpr float a = 0.5f / kL;
    float f4 = 0.25f * u[0];
    float p0 = fminf(fmaxf(f4 - a * u[2], 0.0f), mg);
    float p1 = fminf(fmaxf(f4 + a * u[1], 0.0f), mg);
    float p2 = fminf(fmaxf(f4 + a * u[2], 0.0f), mg);
    float p3 = fminf(fmaxf(f4 - a * u[1], 0.0f), mg);
    out[0] = (p0 + p1) + (p2 + p3);
    out[1] = kL * (p1 - p3);
    out[2] = kL * (p2 - p0);
    out[3] = u[3];
}

// Drone.step, dynamics/quadrotor.py:126-144: integrate with the PREVIOUS limited
// control, clamp attitude, then store the newly limited control.
// integration + attitude clamp with the PREVIOUS limited control (everything of Drone.step but the control hand-over)
template <int INTEG>
__device__ __forceinline__ bool drone_advance(float s[13], const float u_prev[4], const Par &P, float dt)
{
    float inv_m = q_rcp(P.m);
    float k1[13];
    drone_df(s, u_prev, P, inv_m, k1);
    if (INTEG == 0) {
        // RK45 over the frozen RHS of Drone.f (:115-124) == explicit Euler
#pragma unroll
        for (int i = 0; i < 13; ++i) s[i] = fmaf(dt, k1[i], s[i]);
    } else {
        float k2[13], k3[13], k4[13], y[13];
        float h = 0.5f * dt;
#pragma unroll
        for (int i = 0; i < 13; ++i) y[i] = fmaf(h, k1[i], s[i]);
        drone_df(y, u_prev, P, inv_m, k2);
#pragma unroll
        for (int i = 0; i < 13; ++i) y[i] = fmaf(h, k2[i], s[i]);
        drone_df(y, u_prev, P, inv_m, k3);
#pragma unroll
        for (int i = 0; i < 13; ++i) y[i] = fmaf(dt, k3[i], s[i]);
        drone_df(y, u_prev, P, inv_m, k4);
        float d6 = dt * (1.0f / 6.0f);
#pragma unroll
        for (int i = 0; i < 13; ++i) s[i] = fmaf(d6, (k1[i] + k4[i]) + 2.0f * (k2[i] + k3[i]), s[i]);
    }
    return attitude_limit(s);
}

template <int INTEG>
__device__ __forceinline__ bool drone_step(float s[13], float u_prev[4], const float u[4], const Par &P, float dt)
{
    bool over = drone_advance<INTEG>(s, u_prev, P, dt);
    u_limit(u, P.m * kG, u_prev);
    return over;
}

// attitude_controller, controller/PIDController.py:52-74 with w_des = (0, 0, wdz)
__device__ __forceinline__ void attitude_controller(const float qdes[4], float wdz, const float s[13], float M[3])
{
    float dr, dp, dy, nr, np_, ny;
    quat2euler(qdes, dr, dp, dy);
    quat2euler(s + 6, nr, np_, ny);
    M[0] = -10.0f * (dr - nr) + 5.1f * (0.0f - s[10]);
    M[1] = -10.0f * (dp - np_) + 5.1f * (0.0f - s[11]);
    M[2] = -9.5f * (dy - ny) + 4.0f * (wdz - s[12]);
}

// shared tail of hover_controller (:84-102) and vel_controller (:116-134):
// rewrites the desired attitude quaternion in place, returns the thrust.
__device__ __forceinline__ float desired_attitude(float ax, float ay, float az, float qdes[4], float m)
{
    float F = fmaf(m, az, m * kG);
    float sp, cp, sh, ch;
    quat_yaw_trig(qdes, sp, cp, sh, ch);
    constexpr float inv_g = 1.0f / kG;
    float phi_des = (ax * sp - ay * cp) * inv_g;
    float theta_des = (ax * cp + ay * sp) * inv_g;
    float sr, cr, st, ct;
    q_sincos(0.5f * phi_des, sr, cr);
    q_sincos(0.5f * theta_des, st, ct);
    euler2quat_trig(sr, cr, st, ct, sh, ch, qdes);
    return F;
}

// controller.PID (mode 0, PIDController.py:179-185 -> :76-104) or
// controller.vel_controller (mode 1, :106-141) on explicit desired pos / vel.
// dv = state_now.vel - state_last.vel (mode 1; identically 0 inside the envs).
__device__ __forceinline__ void target_control(int mode, const float pdes[3], const float vdes[3], float qdes[4],
                                               float wdz, const float s[13], const float dv[3], float m, float u[4])
{
    float ax, ay, az;
    if (mode == 0) {
        ax = -1.0f * (pdes[0] - s[0]) + -1.65f * (vdes[0] - s[3]);   // :80
        ay = -1.0f * (pdes[1] - s[1]) + -1.65f * (vdes[1] - s[4]);   // :81
        az = 50.0f * (pdes[2] - s[2]) + 8.0f * (vdes[2] - s[5]);     // :82
    } else {
        ax = -0.7f * (vdes[0] - s[3]);                               // :112 (kd_vx = 0)
        ay = -0.7f * (vdes[1] - s[4]);                               // :113 (kd_vy = 0)
        az = 1.0f * (vdes[2] - s[5]) + 0.1f * dv[2];                 // :114
    }
    u[0] = desired_attitude(ax, ay, az, qdes, m);
    attitude_controller(qdes, wdz, s, u + 1);
}

// dock-port states (dynamics/quadrotor.py:213-224) + state2rel (docking_env.py:257-295).
// Ports are (+0.1,0,0) on the chaser and (-0.1,0,0) on the target (docking_env.py:38,51).
// TARGET_LEVEL: the target's attitude is the identity quaternion (every reset state): R_I2B = I folds away.
template <bool TARGET_LEVEL = false>
__device__ __forceinline__ void rel_obs(const float sc[13], const float st[13], float o[12])
{
    Rot A = quat2rot(sc + 6);   // R_I2A
    Rot B = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (!TARGET_LEVEL) B = quat2rot(st + 6);   // R_I2B
    // port offsets in the world frame: R^T port = port_x * (1, r01, r02)
    float bc0 = 0.1f, bc1 = 0.1f * A.r01, bc2 = 0.1f * A.r02;
    float bt0 = -0.1f, bt1 = -0.1f * B.r01, bt2 = -0.1f * B.r02;
    // (p_t - p_c) first: exact-ish in binary32, then the small offsets
    o[0] = (st[0] - sc[0]) + (bt0 - bc0);
    o[1] = (st[1] - sc[1]) + (bt1 - bc1);
    o[2] = (st[2] - sc[2]) + (bt2 - bc2);
    // vel = v + w x b
    float wc0 = sc[10], wc1 = sc[11], wc2 = sc[12];
    float wt0 = st[10], wt1 = st[11], wt2 = st[12];
    float cvx = wc1 * bc2 - wc2 * bc1, cvy = wc2 * bc0 - wc0 * bc2, cvz = wc0 * bc1 - wc1 * bc0;
    float tvx = wt1 * bt2 - wt2 * bt1, tvy = wt2 * bt0 - wt0 * bt2, tvz = wt0 * bt1 - wt1 * bt0;
    o[3] = (st[3] - sc[3]) + (tvx - cvx);
    o[4] = (st[4] - sc[4]) + (tvy - cvy);
    o[5] = (st[5] - sc[5]) + (tvz - cvz);
    // R_A2B = R_I2B @ R_I2A^T  (:267), unit diagonals folded in
    float R00 = 1.0f + B.r01 * A.r01 + B.r02 * A.r02;
    float R01 = A.r10 + B.r01 + B.r02 * A.r12;
    float R02 = A.r20 + B.r01 * A.r21 + B.r02;
    float R10 = B.r10 + A.r01 + B.r12 * A.r02;
    float R11 = B.r10 * A.r10 + 1.0f + B.r12 * A.r12;
    float R12 = B.r10 * A.r20 + A.r21 + B.r12;
    float R20 = B.r20 + B.r21 * A.r01 + A.r02;
    float R21 = B.r20 * A.r10 + B.r21 + A.r12;
    float R22 = B.r20 * A.r20 + B.r21 * A.r21 + 1.0f;
    // rot2euler, utils/transform.py:23-46
    float phi = q_asin(fminf(fmaxf(R12, -1.0f), 1.0f));
    float psi = q_atan2(-R10, R11);
    float th = q_atan2(-R02, R22);
    float theta = (R12 >= 1.0f || R12 < -1.0f) ? 0.0f : th;
    // R_I2B w_B - R_A2B (R_I2A w_A)   (:277; association is immaterial)
    float a0 = wc0 + A.r01 * wc1 + A.r02 * wc2;
    float a1 = A.r10 * wc0 + wc1 + A.r12 * wc2;
    float a2 = A.r20 * wc0 + A.r21 * wc1 + wc2;
    float P = (wt0 + B.r01 * wt1 + B.r02 * wt2) - (R00 * a0 + R01 * a1 + R02 * a2);
    float Q = (B.r10 * wt0 + wt1 + B.r12 * wt2) - (R10 * a0 + R11 * a1 + R12 * a2);
    float Rr = (B.r20 * wt0 + B.r21 * wt1 + wt2) - (R20 * a0 + R21 * a1 + R22 * a2);
    // sin/cos of theta = atan2(-R02, R22) and of phi = asin(R12) by algebra (no sincos)
    bool satur = (R12 >= 1.0f || R12 < -1.0f);
    float h2 = R02 * R02 + R22 * R22;
    float ih = q_rsqrt(h2);
    bool flat = satur || !(h2 > 0.0f);
    float sth = flat ? 0.0f : -R02 * ih;
    float cth = flat ? 1.0f : R22 * ih;
    float sph = fminf(fmaxf(R12, -1.0f), 1.0f);
    float cph = q_sqrt((1.0f - sph) * (1.0f + sph));
    float k = Rr * cth - P * sth;
    float icph = q_rcp(cph);
    o[6] = phi; o[7] = theta; o[8] = psi;
    o[9] = P * cth + Rr * sth;            // :283
    o[10] = Q - (sph * icph) * k;         // :284
    o[11] = k * icph;                     // :285
}

// nominal reset states, docking_env.py:34-57
__device__ __forceinline__ void nominal_init(float sc[13], float st[13])
{
#pragma unroll
    for (int i = 0; i < 13; ++i) { sc[i] = 0.0f; st[i] = 0.0f; }
    sc[0] = 8.0f; sc[1] = -50.0f; sc[2] = 5.0f; sc[6] = 1.0f;
    st[0] = 10.0f; st[1] = -50.0f; st[2] = 5.0f; st[6] = 1.0f;
}

// (0,1] uniform from 32 random bits: one fused multiply-add, pinned identically in the oracle
__device__ __forceinline__ float u01(unsigned v) { return __fmaf_rn((float)v, 2.3283064e-10f, 2.3283064e-10f); }
__device__ __forceinline__ float sym(float u) { return __fmaf_rn(2.0f, u, -1.0f); }
// (0,1) uniform from 16 random bits, (h + 1/2) / 65536: exact in binary32
__device__ __forceinline__ float u16lo(unsigned w) { return __fmaf_rn((float)(w & 0xFFFFu), 1.52587890625e-05f, 7.62939453125e-06f); }
__device__ __forceinline__ float u16hi(unsigned w) { return __fmaf_rn((float)(w >> 16), 1.52587890625e-05f, 7.62939453125e-06f); }
// three (0,1) uniforms from ONE 32-bit word: bits 10..0 and 21..11 on the 11-bit lattice (h + 1/2) / 2048, bits 31..22 on the
// 10-bit lattice (h + 1/2) / 1024; exact in binary32
__device__ __forceinline__ float u11a(unsigned w) { return __fmaf_rn((float)(w & 0x7FFu), 4.8828125e-04f, 2.44140625e-04f); }
__device__ __forceinline__ float u11b(unsigned w) { return __fmaf_rn((float)((w >> 11) & 0x7FFu), 4.8828125e-04f, 2.44140625e-04f); }
__device__ __forceinline__ float u10c(unsigned w) { return __fmaf_rn((float)(w >> 22), 9.765625e-04f, 4.8828125e-04f); }

struct RandCfg {
    uint64_t seed;
    float rr[8];       // pos, vel, euler, rate half-ranges; mass lo,hi; inertia lo,hi
    float par_nom[4];  // nominal mass, Ixx, Iyy, Izz
};

// rocRAND Philox4x32-10 block `blk` of subsequence (stream<<48 | gid)
__device__ __forceinline__ uint4 philox_block(uint64_t seed, uint64_t stream, uint64_t gid, uint64_t blk)
{
    rocrand_state_philox4x32_10 rs;
    rocrand_init(seed, (stream << 48) | gid, 4ull * blk, &rs);
    return rocrand4(&rs);
}

// Randomised initial state (+ per-episode params).  The draw of a possible reset sits in EVERY step of the step kernels (the
// target wave makes it speculatively), so it is kept to what is needed (round 3: the state took two Philox blocks of 16-bit
// uniforms before; a block is ~115 vector instructions, ~8 % of a step):
//   block 2*ctr      the chaser's 12 state uniforms, three per word (u11a, u11b, u10c): word 0 position x,y,z; 1 velocity;
//                    2 euler angles; 3 body rates -- 0.5 m resolved to 0.25-0.5 mm, 0.2 rad to 0.1-0.2 mrad;
//   block 2*ctr + 1  ONLY with per-episode params (QS_RANDOMISE_PARAMS): mass, Ixx (low / high half of word 0), Iyy, Izz (word 1),
//                    16-bit lattice.
// Half-angles stay below pi/4 (qs_create checks init_range[2] <= pi/2), so the reduction-free sincos applies.
// Build extension; the reference's v0/v2 have no randomness.  Pinned bit for bit in the oracle.
template <bool WITH_PAR>
__device__ __forceinline__ void random_init_words(const RandCfg &rc, uint64_t stream, uint64_t gid, uint64_t ctr,
                                                  uint4 &w0, uint4 &w1)
{
    w0 = philox_block(rc.seed, stream, gid, 2ull * ctr + 0);
    if (WITH_PAR) w1 = philox_block(rc.seed, stream, gid, 2ull * ctr + 1);
}

template <bool WITH_PAR>
__device__ __forceinline__ void random_init_apply(const RandCfg &rc, const uint4 &w0, const uint4 &w1, float sc[13],
                                                  float st[13], Par &P)
{
    nominal_init(sc, st);
    sc[0] = __fmaf_rn(sym(u11a(w0.x)), rc.rr[0], 8.0f);
    sc[1] = __fmaf_rn(sym(u11b(w0.x)), rc.rr[0], -50.0f);
    sc[2] = __fmaf_rn(sym(u10c(w0.x)), rc.rr[0], 5.0f);
    sc[3] = sym(u11a(w0.y)) * rc.rr[1];
    sc[4] = sym(u11b(w0.y)) * rc.rr[1];
    sc[5] = sym(u10c(w0.y)) * rc.rr[1];
    float e0 = sym(u11a(w0.z)) * rc.rr[2];
    float e1 = sym(u11b(w0.z)) * rc.rr[2];
    float e2 = sym(u10c(w0.z)) * rc.rr[2];
    float sr, cr, sp, cp, sy, cy;
    q_sincos_small(0.5f * e0, sr, cr);
    q_sincos_small(0.5f * e1, sp, cp);
    q_sincos_small(0.5f * e2, sy, cy);
    euler2quat_trig(sr, cr, sp, cp, sy, cy, sc + 6);
    sc[10] = sym(u11a(w0.w)) * rc.rr[3];
    sc[11] = sym(u11b(w0.w)) * rc.rr[3];
    sc[12] = sym(u10c(w0.w)) * rc.rr[3];
    if (WITH_PAR) {
        P.m = rc.par_nom[0] * __fmaf_rn(rc.rr[5] - rc.rr[4], u16lo(w1.x), rc.rr[4]);
        P.Ixx = rc.par_nom[1] * __fmaf_rn(rc.rr[7] - rc.rr[6], u16hi(w1.x), rc.rr[6]);
        P.Iyy = rc.par_nom[2] * __fmaf_rn(rc.rr[7] - rc.rr[6], u16lo(w1.y), rc.rr[6]);
        P.Izz = rc.par_nom[3] * __fmaf_rn(rc.rr[7] - rc.rr[6], u16hi(w1.y), rc.rr[6]);
    } else {
        P = Par{rc.par_nom[0], rc.par_nom[1], rc.par_nom[2], rc.par_nom[3]};
    }
}

template <bool WITH_PAR>
__device__ __forceinline__ void random_init(const RandCfg &rc, uint64_t stream, uint64_t gid, uint64_t ctr,
                                            float sc[13], float st[13], Par &P)
{
    uint4 w0, w1 = make_uint4(0, 0, 0, 0);
    random_init_words<WITH_PAR>(rc, stream, gid, ctr, w0, w1);
    random_init_apply<WITH_PAR>(rc, w0, w1, sc, st, P);
}

__device__ __forceinline__ void random_action(uint64_t seed, uint64_t gid, uint64_t k, float a[4])
{
    uint4 w = philox_block(seed, STREAM_ACTIONS, gid, k);
    a[0] = sym(u01(w.x)); a[1] = sym(u01(w.y)); a[2] = sym(u01(w.z)); a[3] = sym(u01(w.w));
}

// natural log / tanh on the hardware log2 / exp2 (1 ulp): policy sampling only, never on the env path
__device__ __forceinline__ float q_ln(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
// tanh(x) and 1 - tanh(x)^2 = 4 e / (e + 1)^2, e = exp(2|x|): the second without the cancellation of 1 - t*t
__device__ __forceinline__ float q_tanh(float x, float &sech2)
{
    float e = __builtin_amdgcn_exp2f(fminf(2.0f * fabsf(x), 60.0f) * 1.44269504088896341f);
    float r = q_rcp(e + 1.0f);
    sech2 = 4.0f * (e * r) * r;
    return copysignf(1.0f - 2.0f * r, x);
}

// four standard normals for (env gid, step k): Philox block k of the POLICY stream, Box-Muller on (0,1] uniforms,
// (w.x, w.y) -> n0 = r cos, n1 = r sin; (w.z, w.w) -> n2, n3.  Stands in for tf.random_normal in
// DiagGaussianProbabilityDistribution.sample (rl_baselines/common/distributions.py:426-430); pinned in the oracle.
__device__ __forceinline__ void random_normal4(uint64_t seed, uint64_t gid, uint64_t k, float n[4])
{
    uint4 w = philox_block(seed, STREAM_POLICY, gid, k);
    float r0 = q_sqrt(-2.0f * q_ln(u01(w.x)));
    float r1 = q_sqrt(-2.0f * q_ln(u01(w.z)));
    float s0, c0, s1, c1;
    q_sincos(2.0f * kPi * u01(w.y), s0, c0);
    q_sincos(2.0f * kPi * u01(w.w), s1, c1);
    n[0] = r0 * c0; n[1] = r0 * s0; n[2] = r1 * c1; n[3] = r1 * s1;
}

// per-env registers
struct Env {
    float sc[13], st[13], uc[4], ut[4], qd[4], ls, t;
};

// chaser command: rotor2control @ (std*a + mean), docking_env.py:115 with :98-99 and quadrotor.py:56-59
__device__ __forceinline__ void chaser_command(const float a[4], float m, float u_c[4])
{
    float mean = 0.5f * m * kG;
    float f0 = fmaf(mean, a[0], mean), f1 = fmaf(mean, a[1], mean);
    float f2 = fmaf(mean, a[2], mean), f3 = fmaf(mean, a[3], mean);
    u_c[0] = (f0 + f1) + (f2 + f3);
    u_c[1] = kL * (f1 - f3);
    u_c[2] = kL * (f2 - f0);
    u_c[3] = kLambda * ((f0 - f1) + (f2 - f3));
}

// flags, shaping reward, done: docking_env.py:130-222 (v2: moving_docking_env.py:137-183)
__device__ __forceinline__ void score_step(const float obs[12], const float a[4], float zc, float t, float &ls,
                                           const EnvConst &C, bool lim_c, bool lim_t, float &reward, unsigned &flags)
{
    float np2 = obs[0] * obs[0] + obs[1] * obs[1] + obs[2] * obs[2];
    float nv2 = obs[3] * obs[3] + obs[4] * obs[4] + obs[5] * obs[5];
    float ne2 = obs[6] * obs[6] + obs[7] * obs[7] + obs[8] * obs[8];
    float nr2 = obs[9] * obs[9] + obs[10] * obs[10] + obs[11] * obs[11];
    float na2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    float np_ = q_sqrt(np2), nv = q_sqrt(nv2), ne = q_sqrt(ne2), nr = q_sqrt(nr2), na = q_sqrt(na2);
    bool docked = (np_ < 0.1f) && (nv < 0.1f) && (fabsf(obs[6]) < kLim10) && (fabsf(obs[7]) < kLim10)
                  && (fabsf(obs[8]) < kLim10);                // :130-134
    bool over = (np_ >= C.rmax) || (zc <= 0.1f);              // :141-142
    bool overtime = t >= kTMax;                               // :152
    // shaping, :215-219 / v2 :176-180
    float shaping = -10.0f * np_ * q_rcp(C.rmax) - nv - (10.0f / kPi) * ne - nr - 0.1f * na + (docked ? 1.0f : 0.0f);
    reward = shaping - ls;                                    // :221
    ls = shaping;                                             // :222
    flags = (docked ? FLAG_DOCKED : 0u) | (over ? FLAG_OVERLIMIT : 0u) | (overtime ? FLAG_OVERTIME : 0u)
            | (lim_c ? FLAG_CLIM : 0u) | (lim_t ? FLAG_TLIM : 0u);
}

// DockingEnv.step (docking_env.py:104-231) / MovingDockingEnv.step (moving_docking_env.py:111-192)
template <int INTEG>
__device__ __forceinline__ void env_step(Env &e, const float a[4], const Par &P, const EnvConst &C, float obs[12],
                                         float &reward, unsigned &flags)
{
    e.t += 1.0f;                                              // :108
    float u_c[4], u_t[4];
    chaser_command(a, P.m, u_c);                              // :115
    // target command from the target state BEFORE stepping, :119 / v2 :126
    const float pdes[3] = {10.0f, -50.0f, 5.0f};              // :60
    const float vdes[3] = {C.vdes_x, 0.0f, 0.0f};
    const float dv[3] = {0.0f, 0.0f, 0.0f};                   // state_last aliases state_now (moving_docking_env.py:117)
    target_control(C.kind, pdes, vdes, e.qd, 0.0f, e.st, dv, P.m, u_t);
    bool lim_t = drone_step<INTEG>(e.st, e.ut, u_t, P, C.dt); // :120
    bool lim_c = drone_step<INTEG>(e.sc, e.uc, u_c, P, C.dt); // :121
    rel_obs(e.sc, e.st, obs);                                 // :124-127
    score_step(obs, a, e.sc[2], e.t, e.ls, C, lim_c, lim_t, reward, flags);
}

// DockingEnv.reset (docking_env.py:233-244) + Drone.reset (quadrotor.py:65-78):
// new initial states, stored controls, t and last_shaping zeroed, q_des untouched.
// env_step in two halves for a caller that has the action late (role-split Runner kernel: the target's half runs while the
// policy is still being evaluated).  The target's half does not see the action: its command from the state BEFORE stepping,
// then its step; the chaser's half is the rest.  The same operations on the same operands as env_step, so the same bits.
template <int INTEG>
__device__ __forceinline__ bool env_step_target(Env &e, const Par &P, const EnvConst &C)
{
    float u_t[4];
    const float pdes[3] = {10.0f, -50.0f, 5.0f};              // :60
    const float vdes[3] = {C.vdes_x, 0.0f, 0.0f};
    const float dv[3] = {0.0f, 0.0f, 0.0f};                   // state_last aliases state_now (moving_docking_env.py:117)
    target_control(C.kind, pdes, vdes, e.qd, 0.0f, e.st, dv, P.m, u_t);
    return drone_step<INTEG>(e.st, e.ut, u_t, P, C.dt);       // :120
}

template <int INTEG>
__device__ __forceinline__ void env_step_chaser(Env &e, const float a[4], const Par &P, const EnvConst &C, bool lim_t, float obs[12],
                                                float &reward, unsigned &flags)
{
    e.t += 1.0f;                                              // :108
    float u_c[4];
    chaser_command(a, P.m, u_c);                              // :115
    bool lim_c = drone_step<INTEG>(e.sc, e.uc, u_c, P, C.dt); // :121
    rel_obs(e.sc, e.st, obs);                                 // :124-127
    score_step(obs, a, e.sc[2], e.t, e.ls, C, lim_c, lim_t, reward, flags);
}

template <bool TARGET_LEVEL = false>
__device__ __forceinline__ void env_reset(Env &e, const float ic[13], const float it[13], float obs[12])
{
#pragma unroll
    for (int i = 0; i < 13; ++i) { e.sc[i] = ic[i]; e.st[i] = it[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { e.uc[i] = 0.0f; e.ut[i] = 0.0f; }
    e.ls = 0.0f;
    e.t = 0.0f;
    rel_obs<TARGET_LEVEL>(e.sc, e.st, obs);
}

// HoveringEnv.step, gym-docking/gym_docking/envs/hovering_env.py:47-78: one drone, action in [0,1]^4
// scaled by action_max = m g (:42,:51), reward :62-76, done :68; state_des = hover at (0,0,5), level (:31-35).
template <int INTEG>
__device__ __forceinline__ void hover_step(float s[13], float u_prev[4], const float a[4], const Par &P, float dt,
                                           float &reward, unsigned &flags)
{
    float amax = P.m * kG;
    float f0 = amax * a[0], f1 = amax * a[1], f2 = amax * a[2], f3 = amax * a[3];
    float u[4];
    u[0] = (f0 + f1) + (f2 + f3);
    u[1] = kL * (f1 - f3);
    u[2] = kL * (f2 - f0);
    u[3] = kLambda * ((f0 - f1) + (f2 - f3));
    bool lim = drone_step<INTEG>(s, u_prev, u, P, dt);        // :52
    float pe0 = -s[0], pe1 = -s[1], pe2 = 5.0f - s[2];        // :57
    float r, p, y;
    quat2euler(s + 6, r, p, y);                               // :59 (quat2euler of the level state_des is 0,0,0)
    float npe = q_sqrt(pe0 * pe0 + pe1 * pe1 + pe2 * pe2);
    float nve = q_sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
    float nae = q_sqrt(r * r + p * p + y * y);
    float nwe = q_sqrt(s[10] * s[10] + s[11] * s[11] + s[12] * s[12]);
    bool inside = (npe < 0.1f) && (nve < 0.1f);               // :63
    bool done = (q_sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) > 100.0f) || (nve > 100.0f);   // :68
    float rr = (inside ? 1.0f : 0.0f) + 0.1f - 0.01f * npe - 0.001f * nve - 0.01f * nae - 0.001f * nwe;  // :71-74
    reward = done ? -0.1f : rr;                               // :76
    flags = (inside ? FLAG_DOCKED : 0u) | (done ? FLAG_OVERLIMIT : 0u) | (lim ? FLAG_CLIM : 0u);
}

}  // namespace qs
