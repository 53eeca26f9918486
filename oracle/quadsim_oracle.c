/*
 * quadsim_oracle.c -- CPU restatement of QuadSim's env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity *checker*: it may be
 * imported / linked / executed only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under quadsim_amd/ (the product) may
 * route through it; the product path fails loudly when the HIP library is
 * missing.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors of its
 * own, so this restatement is pinned against outputs of
 * the reference itself, imported in the build container by
 * oracle/gen_goldens.py and committed as fixtures under tests/golden/
 * (G1..G7); tests/test_oracle_vs_golden.py holds it to <= 1e-12 (f64 build).
 *
 * One source, two builds: -DQSO_F32 gives the float instantiation (symbols
 * suffixed _f32), the default is double (suffix _f64).  The f64 build is the
 * oracle; the f32 build shows what plain fp32 evaluation of the same formulas
 * costs in accuracy and serves as the scalar CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout, e.g. dynamics/quadrotor.py).
 *
 * State vector convention (dynamics/quadrotor.py:25):
 *   s[0:3] pos, s[3:6] vel, s[6:10] quaternion (w,x,y,z), s[10:13] body rates.
 * Env record ("rec", 40 reals per env):
 *   [0:13] chaser state, [13:26] target state, [26:30] chaser last limited
 *   control, [30:34] target last limited control, [34:38] target desired
 *   quaternion (PID-mutated, never reset), [38] last_shaping, [39] t.
 * Params ("par", 4 reals per env): mass, Ixx, Iyy, Izz.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef QSO_F32
typedef float real;
#define FN(name) name##_f32
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_TAN tanf
#define R_ASIN asinf
#define R_ATAN2 atan2f
#define R_FABS fabsf
#define R_LOG logf
#else
typedef double real;
#define FN(name) name##_f64
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_TAN tan
#define R_ASIN asin
#define R_ATAN2 atan2
#define R_FABS fabs
#define R_LOG log
#endif

#define QSO_PI 3.14159265358979323846

/* constants: dynamics/quadrotor.py:10-63 */
static const real K_G = (real)9.81;          /* :15 */
static const real K_L = (real)0.086;         /* :20 */
static const double K_KF = 6.11e-8;          /* :43 */
static const double K_KM = 1.5e-9;           /* :44 */

#define REC_SC 0
#define REC_ST 13
#define REC_UC 26
#define REC_UT 30
#define REC_QD 34
#define REC_LS 38
#define REC_T 39
#define REC_LEN 40

/* -------------------------------------------------------------------------
 * utils/transform.py
 * ---------------------------------------------------------------------- */

/* utils/transform.py:143-144 */
static real deg2rad(double deg) { return (real)(deg * QSO_PI / 180.0); }

/* utils/transform.py:4-20 == dynamics/quadrotor.py:226-245.
 * NOT a rotation matrix: `*` is element-wise, qa_hat uses the normalised
 * quaternion, the last term the un-normalised quat[0].  Row-major R[9]. */
void FN(qso_quat2rot)(const real q[4], real R[9])
{
    real nrm = R_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    real n1 = q[1] / nrm, n2 = q[2] / nrm, n3 = q[3] / nrm;
    real h01 = -n3, h02 = n2, h12 = -n1, h10 = n3, h20 = -n2, h21 = n1;
    real w = q[0];
    R[0] = 1;
    R[1] = 2 * h01 * h01 + 2 * w * h01;
    R[2] = 2 * h02 * h02 + 2 * w * h02;
    R[3] = 2 * h10 * h10 + 2 * w * h10;
    R[4] = 1;
    R[5] = 2 * h12 * h12 + 2 * w * h12;
    R[6] = 2 * h20 * h20 + 2 * w * h20;
    R[7] = 2 * h21 * h21 + 2 * w * h21;
    R[8] = 1;
}

/* utils/transform.py:23-46 (Z-X-Y extraction with saturation branches) */
void FN(qso_rot2euler)(const real R[9], real e[3])
{
    real r12 = R[5], r10 = R[3], r11 = R[4], r02 = R[2], r22 = R[8];
    real phi, theta, psi;
    if (r12 < (real)1.0) {
        if (r12 < (real)-1.0) {
            phi = (real)(-QSO_PI / 2.0);
            psi = R_ATAN2(-r10, r11);
            theta = 0;
        } else {
            phi = R_ASIN(r12);
            psi = R_ATAN2(-r10, r11);
            theta = R_ATAN2(-r02, r22);
        }
    } else {
        phi = (real)(QSO_PI / 2.0);
        psi = R_ATAN2(-r10, r11);
        theta = 0;
    }
    e[0] = phi; e[1] = theta; e[2] = psi;
}

/* utils/transform.py:94-120 */
void FN(qso_quat2euler)(const real q[4], real e[3])
{
    real w = q[0], x = q[1], y = q[2], z = q[3];
    real r10 = (real)2.0 * (x * y - w * z);
    real r11 = w * w - x * x + y * y - z * z;
    real r12 = (real)2.0 * (w * x + y * z);
    real r02 = (real)2.0 * (x * z - w * y);
    real r22 = w * w - x * x - y * y + z * z;
    real phi, theta, psi;
    if (r12 < (real)1.0) {
        if (r12 < (real)-1.0) {
            phi = (real)(-QSO_PI / 2.0);
            psi = R_ATAN2(-r10, r11);
            theta = 0;
        } else {
            phi = R_ASIN(r12);
            psi = R_ATAN2(-r10, r11);
            theta = R_ATAN2(-r02, r22);
        }
    } else {
        phi = (real)(QSO_PI / 2.0);
        psi = R_ATAN2(-r10, r11);
        theta = 0;
    }
    e[0] = phi; e[1] = theta; e[2] = psi;
}

/* utils/transform.py:123-136 (standard ZYX half-angle; NOT the inverse of
 * quat2euler above) */
void FN(qso_euler2quat)(const real e[3], real q[4])
{
    real cy = R_COS(e[2] * (real)0.5), sy = R_SIN(e[2] * (real)0.5);
    real cp = R_COS(e[1] * (real)0.5), sp = R_SIN(e[1] * (real)0.5);
    real cr = R_COS(e[0] * (real)0.5), sr = R_SIN(e[0] * (real)0.5);
    q[0] = cr * cp * cy - sr * sp * sy;
    q[1] = sr * cp * cy - cr * sp * sy;
    q[2] = sr * cp * sy + cr * sp * cy;
    q[3] = cr * cp * sy + sr * sp * cy;
}

/* -------------------------------------------------------------------------
 * dynamics/quadrotor.py
 * ---------------------------------------------------------------------- */

/* Drone.df, dynamics/quadrotor.py:80-113 */
void FN(qso_drone_df)(const real s[13], const real u[4], const real par[4], real ds[13])
{
    real mass = par[0], Ixx = par[1], Iyy = par[2], Izz = par[3];
    real F = u[0];                                            /* :82 */
    real M0 = Ixx * u[1], M1 = Iyy * u[2], M2 = Izz * u[3];   /* :83 */
    const real *q = s + 6, *w = s + 10;
    real R[9];
    FN(qso_quat2rot)(q, R);                                   /* :90 */
    /* R_b2w @ [0,0,F] = F * column 2 of R^T = F * row 2 of R  (:91-94) */
    real inv_m = (real)1.0 / mass;
    ds[0] = s[3]; ds[1] = s[4]; ds[2] = s[5];                 /* :108 */
    ds[3] = inv_m * (R[6] * F - 0);
    ds[4] = inv_m * (R[7] * F - 0);
    ds[5] = inv_m * (R[8] * F - mass * K_G);                  /* :94 */
    real e_quat = (real)1.0 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); /* :97 */
    real k0 = -w[0] * q[1] - w[1] * q[2] - w[2] * q[3];       /* :98 */
    real k1 = w[0] * q[0] - w[1] * q[2] + w[2] * q[3];        /* :99 */
    real k2 = w[1] * q[0] + w[2] * q[1] - w[0] * q[3];        /* :100 */
    real k3 = w[2] * q[0] - w[1] * q[1] + w[0] * q[2];        /* :101 */
    real kq = (real)2.0 * e_quat;                             /* :96,:103 */
    ds[6] = (real)-0.5 * k0 + kq * q[0];
    ds[7] = (real)-0.5 * k1 + kq * q[1];
    ds[8] = (real)-0.5 * k2 + kq * q[2];
    ds[9] = (real)-0.5 * k3 + kq * q[3];
    /* :105  inv(I) @ (M - w x (I w)) with diagonal I */
    real Iw0 = Ixx * w[0], Iw1 = Iyy * w[1], Iw2 = Izz * w[2];
    real c0 = w[1] * Iw2 - w[2] * Iw1;
    real c1 = w[2] * Iw0 - w[0] * Iw2;
    real c2 = w[0] * Iw1 - w[1] * Iw0;
    ds[10] = (M0 - c0) / Ixx;
    ds[11] = (M1 - c1) / Iyy;
    ds[12] = (M2 - c2) / Izz;
}

/* Drone.attitude_limit + the write-back in Drone.step,
 * dynamics/quadrotor.py:146-168 and :135-138.  Returns the over-limit flag. */
int FN(qso_attitude_limit)(real s[13])
{
    real e[3], lim[4], tmp[3];
    int over = 0;
    FN(qso_quat2euler)(s + 6, e);                             /* :152 */
    real r = e[0], p = e[1], y = e[2];
    real l85 = deg2rad(85.0), l175 = deg2rad(175.0);
    if (R_FABS(r) >= l85) {                                   /* :156 */
        tmp[0] = (r > 0 ? l85 : (r < 0 ? -l85 : 0)); tmp[1] = p; tmp[2] = y;
        FN(qso_euler2quat)(tmp, lim); over = 1;
    }
    if (R_FABS(p) >= l85) {                                   /* :159 overrides */
        tmp[0] = r; tmp[1] = (p > 0 ? l85 : (p < 0 ? -l85 : 0)); tmp[2] = y;
        FN(qso_euler2quat)(tmp, lim); over = 1;
    }
    if (R_FABS(y) >= l175) {                                  /* :162 overrides */
        tmp[0] = r; tmp[1] = p; tmp[2] = (y > 0 ? l175 : (y < 0 ? -l175 : 0));
        FN(qso_euler2quat)(tmp, lim); over = 1;
    }
    if (R_FABS(r) <= l85 && R_FABS(p) <= l85 && R_FABS(y) <= l175) /* :165 */
        over = 0;
    if (over) {                                               /* :136-138 */
        s[6] = lim[0]; s[7] = lim[1]; s[8] = lim[2]; s[9] = lim[3];
        s[10] = 0; s[11] = 0; s[12] = 0;
    }
    return over;
}

/* Drone.u_limit, dynamics/quadrotor.py:171-187 (A :47-50, B :52-54).
 * Yaw is not part of the mix; u[3] passes through. */
void FN(qso_u_limit)(const real u[4], const real par[4], real out[4])
{
    real mass = par[0];
    real hi = ((real)4.0 * mass * K_G) / (real)4.0;           /* F_max/4, :22,:179 */
    real lo = 0;                                              /* F_min/4, :23,:180 */
    real a = (real)0.5 / K_L;
    real p0 = (real)0.25 * u[0] + (-a) * u[2];
    real p1 = (real)0.25 * u[0] + a * u[1];
    real p2 = (real)0.25 * u[0] + a * u[2];
    real p3 = (real)0.25 * u[0] + (-a) * u[1];
    if (p0 > hi) p0 = hi;
    if (p0 < lo) p0 = lo;
    if (p1 > hi) p1 = hi;
    if (p1 < lo) p1 = lo;
    if (p2 > hi) p2 = hi;
    if (p2 < lo) p2 = lo;
    if (p3 > hi) p3 = hi;
    if (p3 < lo) p3 = lo;
    out[0] = p0 + p1 + p2 + p3;                               /* :182 */
    out[1] = K_L * p1 - K_L * p3;                             /* :185 row 1 of B */
    out[2] = -K_L * p0 + K_L * p2;                            /* :185 row 2 of B */
    out[3] = u[3];                                            /* :186 */
}

/* Drone.step, dynamics/quadrotor.py:126-144.
 * integ 0 ("frozen"): what the reference actually computes -- Drone.f (:115-124)
 * ignores (t, y), so every RK45 stage returns df(state, u_prev) and the step
 * is state + dt*df(state, u_prev).
 * integ 1 ("rk4"): classic RK4 re-evaluating df at the stage states, u_prev
 * held over the step.  The reference has no counterpart: parity unpinned for
 * this mode (it is checked HIP-vs-oracle only).
 * The control applied is the PREVIOUS limited control (one-step delay, :132-140). */
int FN(qso_drone_step)(real s[13], real u_prev[4], const real u[4], const real par[4], real dt, int integ)
{
    real k1[13];
    int i, over;
    FN(qso_drone_df)(s, u_prev, par, k1);
    if (integ == 0) {
        for (i = 0; i < 13; ++i) s[i] = s[i] + dt * k1[i];
    } else {
        real k2[13], k3[13], k4[13], y[13];
        for (i = 0; i < 13; ++i) y[i] = s[i] + (real)0.5 * dt * k1[i];
        FN(qso_drone_df)(y, u_prev, par, k2);
        for (i = 0; i < 13; ++i) y[i] = s[i] + (real)0.5 * dt * k2[i];
        FN(qso_drone_df)(y, u_prev, par, k3);
        for (i = 0; i < 13; ++i) y[i] = s[i] + dt * k3[i];
        FN(qso_drone_df)(y, u_prev, par, k4);
        for (i = 0; i < 13; ++i)
            s[i] = s[i] + (dt / (real)6.0) * (k1[i] + (real)2.0 * k2[i] + (real)2.0 * k3[i] + k4[i]);
    }
    over = FN(qso_attitude_limit)(s);                         /* :135-138 */
    FN(qso_u_limit)(u, par, u_prev);                          /* :140 */
    return over;
}

/* Drone.get_dock_port_state, dynamics/quadrotor.py:213-224 ('pos','vel' only:
 * 'quat' is computed by the reference but never consumed by state2rel). */
void FN(qso_dock_port)(const real s[13], const real port[3], real pos[3], real vel[3])
{
    real R[9], b[3];
    FN(qso_quat2rot)(s + 6, R);
    /* R_b2w @ port = R^T port */
    b[0] = R[0] * port[0] + R[3] * port[1] + R[6] * port[2];
    b[1] = R[1] * port[0] + R[4] * port[1] + R[7] * port[2];
    b[2] = R[2] * port[0] + R[5] * port[1] + R[8] * port[2];
    pos[0] = s[0] + b[0]; pos[1] = s[1] + b[1]; pos[2] = s[2] + b[2];
    /* w_sk @ b, :219-222 */
    real w0 = s[10], w1 = s[11], w2 = s[12];
    vel[0] = s[3] + (0 * b[0] + (-w2) * b[1] + w1 * b[2]);
    vel[1] = s[4] + (w2 * b[0] + 0 * b[1] + (-w0) * b[2]);
    vel[2] = s[5] + ((-w1) * b[0] + w0 * b[1] + 0 * b[2]);
}

/* -------------------------------------------------------------------------
 * controller/PIDController.py
 * ---------------------------------------------------------------------- */

/* attitude_controller, controller/PIDController.py:52-74 (gains :11-18) */
static void attitude_controller(const real sdes[13], const real s[13], real M[3])
{
    real ed[3], en[3];
    FN(qso_quat2euler)(sdes + 6, ed);                         /* :61 */
    FN(qso_quat2euler)(s + 6, en);                            /* :62 */
    real e0 = ed[0] - en[0], e1 = ed[1] - en[1], e2 = ed[2] - en[2];
    real w0 = sdes[10] - s[10], w1 = sdes[11] - s[11], w2 = sdes[12] - s[12];
    M[0] = (real)-10.0 * e0 + (real)5.1 * w0;                 /* :69 */
    M[1] = (real)-10.0 * e1 + (real)5.1 * w1;                 /* :70 */
    M[2] = (real)0.0 + (real)-9.5 * e2 + (real)4.0 * w2;      /* :71 */
}

/* shared tail of hover_controller / vel_controller,
 * controller/PIDController.py:84-102 == :116-134: MUTATES sdes[6:12]. */
static real desired_attitude(const real acc[3], real sdes[13], real mass)
{
    real F = mass * K_G + mass * acc[2];                      /* :84 */
    real att[3], q[4];
    FN(qso_quat2euler)(sdes + 6, att);                        /* :87 */
    real psi = att[2];
    real phi_des = (acc[0] * R_SIN(psi) - acc[1] * R_COS(psi)) / K_G;   /* :90 */
    real theta_des = (acc[0] * R_COS(psi) + acc[1] * R_SIN(psi)) / K_G; /* :91 */
    att[0] = phi_des; att[1] = theta_des; att[2] = psi;
    FN(qso_euler2quat)(att, q);                               /* :100 */
    sdes[6] = q[0]; sdes[7] = q[1]; sdes[8] = q[2]; sdes[9] = q[3];
    sdes[10] = 0; sdes[11] = 0;                               /* :101-102 */
    return F;
}

/* controller.PID, controller/PIDController.py:179-185 -> hover_controller
 * :76-104 -> attitude_controller :52-74.  sdes is read AND written. */
void FN(qso_ctrl_pid)(real sdes[13], const real s[13], real mass, real u[4])
{
    real acc[3];
    acc[0] = (real)-1.0 * (sdes[0] - s[0]) + (real)-1.65 * (sdes[3] - s[3]);  /* :80 */
    acc[1] = (real)-1.0 * (sdes[1] - s[1]) + (real)-1.65 * (sdes[4] - s[4]);  /* :81 */
    acc[2] = (real)50.0 * (sdes[2] - s[2]) + (real)8.0 * (sdes[5] - s[5]);    /* :82 */
    u[0] = desired_attitude(acc, sdes, mass);
    attitude_controller(sdes, s, u + 1);
}

/* controller.vel_controller, controller/PIDController.py:106-141.
 * s_last is the `state_last` argument (e_dv = s - s_last, :110). */
void FN(qso_ctrl_vel)(real sdes[13], const real s[13], const real s_last[13], real mass, real u[4])
{
    real acc[3];
    acc[0] = (real)-0.7 * (sdes[3] - s[3]) + (real)0.0 * (s[3] - s_last[3]);  /* :112 */
    acc[1] = (real)-0.7 * (sdes[4] - s[4]) + (real)0.0 * (s[4] - s_last[4]);  /* :113 */
    acc[2] = (real)1.0 * (sdes[5] - s[5]) + (real)0.1 * (s[5] - s_last[5]);   /* :114 */
    u[0] = desired_attitude(acc, sdes, mass);
    attitude_controller(sdes, s, u + 1);
}

/* -------------------------------------------------------------------------
 * gym-docking/gym_docking/envs/docking_env.py (v0) and moving_docking_env.py (v2)
 * ---------------------------------------------------------------------- */

/* state2rel, docking_env.py:257-295 (byte-identical at moving_docking_env.py:222-260) */
void FN(qso_state2rel)(const real sc[13], const real st[13],
                       const real cpos[3], const real cvel[3],
                       const real tpos[3], const real tvel[3], real o[12])
{
    real RB[9], RA[9], RAB[9], RABA[9], e[3];
    int i, j, k;
    FN(qso_quat2rot)(st + 6, RB);                             /* :258 R_I2B */
    FN(qso_quat2rot)(sc + 6, RA);                             /* :259 R_I2A */
    for (i = 0; i < 3; ++i) { o[i] = tpos[i] - cpos[i]; o[3 + i] = tvel[i] - cvel[i]; } /* :263-264 */
    /* R_A2B = R_I2B @ R_I2A^T, :267 */
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
            real acc = 0;
            for (k = 0; k < 3; ++k) acc += RB[3 * i + k] * RA[3 * j + k];
            RAB[3 * i + j] = acc;
        }
    FN(qso_rot2euler)(RAB, e);                                /* :269 */
    /* (R_A2B @ R_I2A) @ omega_A -- left-associated, :277 */
    for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
            real acc = 0;
            for (k = 0; k < 3; ++k) acc += RAB[3 * i + k] * RA[3 * k + j];
            RABA[3 * i + j] = acc;
        }
    real rel[3];
    for (i = 0; i < 3; ++i) {
        real a = 0, b = 0;
        for (k = 0; k < 3; ++k) { a += RB[3 * i + k] * st[10 + k]; b += RABA[3 * i + k] * sc[10 + k]; }
        rel[i] = a - b;
    }
    real P = rel[0], Q = rel[1], Rr = rel[2];
    real phi = e[0], theta = e[1];
    real ct = R_COS(theta), sth = R_SIN(theta);
    o[6] = e[0]; o[7] = e[1]; o[8] = e[2];
    o[9] = P * ct + Rr * sth;                                 /* :283 */
    o[10] = Q - R_TAN(phi) * (Rr * ct - P * sth);             /* :284 */
    o[11] = (Rr * ct - P * sth) / R_COS(phi);                 /* :285 */
}

static const real PORT_C[3] = {(real)0.1, 0, 0};   /* docking_env.py:38 */
static const real PORT_T[3] = {(real)-0.1, 0, 0};  /* docking_env.py:51 */

static void rel_obs(const real sc[13], const real st[13], real o[12])
{
    real cp[3], cv[3], tp[3], tv[3];
    FN(qso_dock_port)(sc, PORT_C, cp, cv);
    FN(qso_dock_port)(st, PORT_T, tp, tv);
    FN(qso_state2rel)(sc, st, cp, cv, tp, tv, o);
}

/* observation of an arbitrary (chaser, target) state pair: what reset() returns,
 * docking_env.py:236-238 */
void FN(qso_rel_obs)(const real sc[13], const real st[13], real o[12]) { rel_obs(sc, st, o); }

/* nominal reset states, docking_env.py:34-57 */
void FN(qso_nominal_init)(real sc[13], real st[13])
{
    memset(sc, 0, 13 * sizeof(real)); memset(st, 0, 13 * sizeof(real));
    sc[0] = 8; sc[1] = -50; sc[2] = 5; sc[6] = 1;
    st[0] = 10; st[1] = -50; st[2] = 5; st[6] = 1;
}

/* fresh env record as left by DockingEnv.__init__ (docking_env.py:15-102) */
void FN(qso_env_init)(real rec[REC_LEN])
{
    memset(rec, 0, REC_LEN * sizeof(real));
    FN(qso_nominal_init)(rec + REC_SC, rec + REC_ST);
    rec[REC_QD] = 1;                                          /* :61,:64 */
}

/* DockingEnv.reset, docking_env.py:233-244 with Drone.reset quadrotor.py:65-78:
 * restores the given initial states, zeroes both stored controls, t and
 * last_shaping; does NOT touch target_state_des (rec[REC_QD..]). */
void FN(qso_env_reset)(real rec[REC_LEN], const real init_c[13], const real init_t[13], real obs[12])
{
    int i;
    for (i = 0; i < 13; ++i) { rec[REC_SC + i] = init_c[i]; rec[REC_ST + i] = init_t[i]; }
    for (i = 0; i < 8; ++i) rec[REC_UC + i] = 0;              /* quadrotor.py:76 */
    rec[REC_LS] = 0; rec[REC_T] = 0;                          /* docking_env.py:240-243 */
    rel_obs(rec + REC_SC, rec + REC_ST, obs);
}

/* DockingEnv.step docking_env.py:104-231 (kind 0) /
 * MovingDockingEnv.step moving_docking_env.py:111-192 (kind 1).
 * flags: bit0 flag_docking, bit1 done_overlimit, bit2 done_overtime,
 *        bit3 chaser attitude limiter fired, bit4 target limiter fired
 *        (bits 3-4 are diagnostics of ours, used by tests to find knife-edges). */
void FN(qso_env_step)(real rec[REC_LEN], const real a[4], const real par[4], int kind,
                      real dt, int integ, real obs[12], real *reward, int *done, int *flags)
{
    real *sc = rec + REC_SC, *st = rec + REC_ST, *uc = rec + REC_UC, *ut = rec + REC_UT;
    real mass = par[0];
    real lambda = (real)(K_KM / K_KF);                        /* quadrotor.py:45 */
    real sdes[13], u_c[4], u_t[4];
    int i;

    rec[REC_T] += 1;                                          /* docking_env.py:108 */

    /* chaser command, docking_env.py:115 with :98-99 and quadrotor.py:56-59 */
    real mean = (real)1.0 * mass * K_G / (real)2.0;
    real f0 = mean * a[0] + mean, f1 = mean * a[1] + mean, f2 = mean * a[2] + mean, f3 = mean * a[3] + mean;
    u_c[0] = f0 + f1 + f2 + f3;
    u_c[1] = 0 * f0 + K_L * f1 + 0 * f2 + (-K_L) * f3;
    u_c[2] = (-K_L) * f0 + 0 * f1 + K_L * f2 + 0 * f3;
    u_c[3] = lambda * f0 + (-lambda) * f1 + lambda * f2 + (-lambda) * f3;

    /* target command from the target state BEFORE stepping, :119 / v2 :126 */
    memset(sdes, 0, sizeof sdes);
    sdes[0] = 10; sdes[1] = -50; sdes[2] = 5;                 /* docking_env.py:60-63 */
    if (kind == 1) sdes[3] = (real)0.2;                       /* moving_docking_env.py:62,65 */
    for (i = 0; i < 4; ++i) sdes[6 + i] = rec[REC_QD + i];
    if (kind == 0)
        FN(qso_ctrl_pid)(sdes, st, mass, u_t);
    else
        FN(qso_ctrl_vel)(sdes, st, st, mass, u_t);            /* state_last aliases state_now: e_dv == 0 */
    for (i = 0; i < 4; ++i) rec[REC_QD + i] = sdes[6 + i];

    int lim_t = FN(qso_drone_step)(st, ut, u_t, par, dt, integ);   /* :120 */
    int lim_c = FN(qso_drone_step)(sc, uc, u_c, par, dt, integ);   /* :121 */

    rel_obs(sc, st, obs);                                     /* :124-127 */

    real np_ = R_SQRT(obs[0] * obs[0] + obs[1] * obs[1] + obs[2] * obs[2]);
    real nv = R_SQRT(obs[3] * obs[3] + obs[4] * obs[4] + obs[5] * obs[5]);
    real l10 = deg2rad(10.0);
    int docked = (np_ < (real)0.1) && (nv < (real)0.1) && (R_FABS(obs[6]) < l10)
                 && (R_FABS(obs[7]) < l10) && (R_FABS(obs[8]) < l10);      /* :130-134 */
    real rmax = (kind == 0) ? (real)3.0 : (real)10.0;         /* :141 / v2 :148 */
    int over = (np_ >= rmax) || (sc[2] <= (real)0.1);         /* :141-142 */
    int overtime = (rec[REC_T] >= (real)600.0);               /* :152 */
    *done = over || overtime;                                 /* :155 */

    real ra = R_SQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]);  /* :166 */
    /* shaping, :215-219 / v2 :176-180 */
    real q0 = obs[0] / rmax, q1 = obs[1] / rmax, q2 = obs[2] / rmax;
    real pi_r = (real)QSO_PI;
    real e0 = obs[6] / pi_r, e1 = obs[7] / pi_r, e2 = obs[8] / pi_r;
    real shaping = (real)-10.0 * R_SQRT(q0 * q0 + q1 * q1 + q2 * q2)
                   - (real)1.0 * nv
                   - (real)10.0 * R_SQRT(e0 * e0 + e1 * e1 + e2 * e2)
                   - (real)1.0 * R_SQRT(obs[9] * obs[9] + obs[10] * obs[10] + obs[11] * obs[11])
                   - (real)0.1 * ra + (real)1.0 * (docked ? (real)1.0 : (real)0.0);
    *reward = shaping - rec[REC_LS];                          /* :221 */
    rec[REC_LS] = shaping;                                    /* :222 */
    *flags = (docked ? 1 : 0) | (over ? 2 : 0) | (overtime ? 4 : 0) | (lim_c ? 8 : 0) | (lim_t ? 16 : 0);
}

/* -------------------------------------------------------------------------
 * Philox4x32-10 as rocRAND's device engine evaluates it
 * (rocrand/rocrand_philox4x32_10.h, ROCm 7.2; Random123 constants).  Integer
 * work: the HIP reset kernel must match this bit for bit.
 * counter = (offset/4 as 64 bit, subsequence as 64 bit), key = seed.
 * Only offsets that are multiples of 4 are used by the build.
 * ---------------------------------------------------------------------- */
void FN(qso_philox4x32_10)(uint64_t seed, uint64_t subsequence, uint64_t block, uint32_t out[4])
{
    uint32_t c0 = (uint32_t)block, c1 = (uint32_t)(block >> 32);
    uint32_t c2 = (uint32_t)subsequence, c3 = (uint32_t)(subsequence >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int r;
    for (r = 0; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uint32 -> (0,1] float exactly as the HIP side does it: one fused
 * multiply-add in binary32, 2^-32 * v + 2^-32 (rocrand_uniform.h:65-68 writes
 * the same expression unfused; we pin the fused form on both sides). */
float FN(qso_u01)(uint32_t v)
{
    return fmaf((float)v, 2.3283064e-10f, 2.3283064e-10f);
}

/* RNG stream ids (upper 16 bits of the Philox subsequence) */
#define STREAM_AUTORESET 0ull
#define STREAM_RESET 1ull
#define STREAM_ACTIONS 2ull
#define STREAM_POLICY 4ull

/* (0,1) uniform from 16 random bits: (h + 1/2) / 65536, exact in binary32 */
static float u16f(uint32_t h) { return fmaf((float)h, 1.52587890625e-05f, 7.62939453125e-06f); }

/* Randomised initial state + per-episode params for env `gid` (global env id).
 * This is the build's own extension (the reference has no randomness in
 * v0/v2; the ranges are the commented-out lines docking_env.py:34-37).
 * rr[0..3] = half-ranges for chaser pos / vel / euler / body rates,
 * rr[4..5] = mass scale [lo,hi], rr[6..7] = inertia scale [lo,hi].
 * Philox block 2*ctr of subsequence (stream<<48 | gid) carries the 12 state
 * uniforms, three per 32-bit word: bits 10..0 and 21..11 on the 11-bit lattice
 * (h + 1/2) / 2048, bits 31..22 on the 10-bit lattice (h + 1/2) / 1024 -- word
 * 0 position, 1 velocity, 2 euler angles, 3 body rates; block 2*ctr + 1 carries
 * the 4 per-episode params as 16-bit uniforms (mass, Ixx = low / high half of
 * word 0; Iyy, Izz = word 1).  (Round 3: the state draw was two blocks of 16-bit
 * uniforms; the device makes this draw speculatively in every step.)
 * u16[0..11] = the state uniforms, u16[12..15] = the params uniforms.
 * All arithmetic in binary32 with explicit fmaf so that HIP == CPU bit for bit
 * up to the euler2quat sin/cos (compared to tolerance). */
void FN(qso_random_init)(uint64_t seed, uint64_t stream, uint64_t gid, uint64_t ctr,
                         const float rr[8], const float par_nom[4],
                         float sc[13], float st[13], float par[4], float u16[16])
{
    uint32_t w[8];
    float u[16];
    int i;
    for (i = 0; i < 2; ++i)
        FN(qso_philox4x32_10)(seed, (stream << 48) | gid, 2ull * ctr + (uint64_t)i, w + 4 * i);
    for (i = 0; i < 4; ++i) {
        u[3 * i + 0] = fmaf((float)(w[i] & 0x7FFu), 4.8828125e-04f, 2.44140625e-04f);
        u[3 * i + 1] = fmaf((float)((w[i] >> 11) & 0x7FFu), 4.8828125e-04f, 2.44140625e-04f);
        u[3 * i + 2] = fmaf((float)(w[i] >> 22), 9.765625e-04f, 4.8828125e-04f);
    }
    u[12] = u16f(w[4] & 0xFFFFu); u[13] = u16f(w[4] >> 16);
    u[14] = u16f(w[5] & 0xFFFFu); u[15] = u16f(w[5] >> 16);
    if (u16) for (i = 0; i < 16; ++i) u16[i] = u[i];
    float e[3];
    for (i = 0; i < 13; ++i) { sc[i] = 0; st[i] = 0; }
    sc[0] = fmaf(fmaf(2.0f, u[0], -1.0f), rr[0], 8.0f);
    sc[1] = fmaf(fmaf(2.0f, u[1], -1.0f), rr[0], -50.0f);
    sc[2] = fmaf(fmaf(2.0f, u[2], -1.0f), rr[0], 5.0f);
    sc[3] = fmaf(2.0f, u[3], -1.0f) * rr[1];
    sc[4] = fmaf(2.0f, u[4], -1.0f) * rr[1];
    sc[5] = fmaf(2.0f, u[5], -1.0f) * rr[1];
    e[0] = fmaf(2.0f, u[6], -1.0f) * rr[2];
    e[1] = fmaf(2.0f, u[7], -1.0f) * rr[2];
    e[2] = fmaf(2.0f, u[8], -1.0f) * rr[2];
    {
        float cy = cosf(e[2] * 0.5f), sy = sinf(e[2] * 0.5f);
        float cp = cosf(e[1] * 0.5f), sp = sinf(e[1] * 0.5f);
        float cr = cosf(e[0] * 0.5f), sr = sinf(e[0] * 0.5f);
        sc[6] = cr * cp * cy - sr * sp * sy;
        sc[7] = sr * cp * cy - cr * sp * sy;
        sc[8] = sr * cp * sy + cr * sp * cy;
        sc[9] = cr * cp * sy + sr * sp * cy;
    }
    sc[10] = fmaf(2.0f, u[9], -1.0f) * rr[3];
    sc[11] = fmaf(2.0f, u[10], -1.0f) * rr[3];
    sc[12] = fmaf(2.0f, u[11], -1.0f) * rr[3];
    st[0] = 10.0f; st[1] = -50.0f; st[2] = 5.0f; st[6] = 1.0f;
    par[0] = par_nom[0] * fmaf(rr[5] - rr[4], u[12], rr[4]);
    par[1] = par_nom[1] * fmaf(rr[7] - rr[6], u[13], rr[6]);
    par[2] = par_nom[2] * fmaf(rr[7] - rr[6], u[14], rr[6]);
    par[3] = par_nom[3] * fmaf(rr[7] - rr[6], u[15], rr[6]);
}

/* synthetic U(-1,1) action for (env gid, step k): Philox block k of the action
 * stream; a = 2u-1 (one fmaf). */
void FN(qso_random_action)(uint64_t seed, uint64_t gid, uint64_t k, float a[4])
{
    uint32_t w[4];
    int i;
    FN(qso_philox4x32_10)(seed, (STREAM_ACTIONS << 48) | gid, k, w);
    for (i = 0; i < 4; ++i) a[i] = fmaf(2.0f, FN(qso_u01)(w[i]), -1.0f);
}

/* four standard normals for (env gid, step k): Philox block k of the POLICY
 * stream, Box-Muller on the (0,1] uniforms: (w0,w1) -> r cos, r sin; (w2,w3)
 * likewise.  What the build draws in place of tf.random_normal in
 * DiagGaussianProbabilityDistribution.sample (rl_baselines/common/distributions.py:426-430). */
void FN(qso_normal4)(uint64_t seed, uint64_t gid, uint64_t k, real n[4])
{
    uint32_t w[4];
    real r0, r1, a0, a1;
    FN(qso_philox4x32_10)(seed, (STREAM_POLICY << 48) | gid, k, w);
    r0 = R_SQRT((real)-2 * R_LOG((real)FN(qso_u01)(w[0])));
    r1 = R_SQRT((real)-2 * R_LOG((real)FN(qso_u01)(w[2])));
    a0 = (real)(2.0 * QSO_PI) * (real)FN(qso_u01)(w[1]);
    a1 = (real)(2.0 * QSO_PI) * (real)FN(qso_u01)(w[3]);
    n[0] = r0 * R_COS(a0); n[1] = r0 * R_SIN(a0); n[2] = r1 * R_COS(a1); n[3] = r1 * R_SIN(a1);
}

/* -------------------------------------------------------------------------
 * Vectorised driver with SB2-VecEnv auto-reset semantics (the worker loop of
 * SubprocVecEnv, contract visible at rl_baselines/ppo2/ppo2.py:472-499):
 * on done the terminal observation is kept, the env is reset and the returned
 * observation is the first one of the new episode.
 *   rec[N][40], par[N][4] (updated on reset when randomise != 0), actions[N][4],
 *   obs[N][12], reward[N], done[N], flags[N], term_obs[N][12] (nullable; rows
 *   written only where done).
 *   randomise: 0 nominal reset, 1 random init (rr[0..3]), 2 also params (rr[4..7])
 *   step_idx: global step counter k of this call (auto-reset draws use ctr k+1)
 *   gid0: global id of env 0 (multi-GPU shard offset).
 * ---------------------------------------------------------------------- */
void FN(qso_vec_step)(int64_t N, real *rec, real *par, const real *actions, int kind, real dt, int integ,
                      int auto_reset, int randomise, uint64_t seed, uint64_t step_idx, uint64_t gid0,
                      const float rr[8], const float par_nom[4],
                      real *obs, real *reward, uint8_t *done, uint8_t *flags, real *term_obs)
{
    int64_t i;
    for (i = 0; i < N; ++i) {
        real *r = rec + i * REC_LEN;
        real rew; int d, f, j;
        FN(qso_env_step)(r, actions + 4 * i, par + 4 * i, kind, dt, integ, obs + 12 * i, &rew, &d, &f);
        reward[i] = rew; done[i] = (uint8_t)d; flags[i] = (uint8_t)f;
        if (d && auto_reset) {
            real ic[13], it[13];
            if (term_obs) for (j = 0; j < 12; ++j) term_obs[12 * i + j] = obs[12 * i + j];
            if (randomise) {
                float fc[13], ft[13], fp[4];
                FN(qso_random_init)(seed, STREAM_AUTORESET, gid0 + (uint64_t)i, step_idx + 1, rr, par_nom, fc, ft, fp, 0);
                for (j = 0; j < 13; ++j) { ic[j] = fc[j]; it[j] = ft[j]; }
                if (randomise >= 2) for (j = 0; j < 4; ++j) par[4 * i + j] = fp[j];
            } else {
                FN(qso_nominal_init)(ic, it);
            }
            FN(qso_env_reset)(r, ic, it, obs + 12 * i);
        }
    }
}

/* explicit (masked) reset of the vector env; mask nullable = all.
 * Uses the RESET stream with ctr = step_idx. */
void FN(qso_vec_reset)(int64_t N, real *rec, real *par, const uint8_t *mask, int randomise,
                       uint64_t seed, uint64_t step_idx, uint64_t gid0,
                       const float rr[8], const float par_nom[4], real *obs)
{
    int64_t i; int j;
    for (i = 0; i < N; ++i) {
        real ic[13], it[13];
        if (mask && !mask[i]) continue;
        if (randomise) {
            float fc[13], ft[13], fp[4];
            FN(qso_random_init)(seed, STREAM_RESET, gid0 + (uint64_t)i, step_idx, rr, par_nom, fc, ft, fp, 0);
            for (j = 0; j < 13; ++j) { ic[j] = fc[j]; it[j] = ft[j]; }
            if (randomise >= 2) for (j = 0; j < 4; ++j) par[4 * i + j] = fp[j];
        } else {
            FN(qso_nominal_init)(ic, it);
        }
        FN(qso_env_reset)(rec + i * REC_LEN, ic, it, obs + 12 * i);
    }
}

/* T-step rollout driver == T calls of qso_vec_step with actions[t] and
 * outputs stacked [T][N][...]; the CPU twin of the fused HIP rollout kernel
 * and of the loop body at rl_baselines/ppo2/ppo2.py:472-499. */
void FN(qso_vec_rollout)(int64_t T, int64_t N, real *rec, real *par, const real *actions, int kind, real dt,
                         int integ, int randomise, uint64_t seed, uint64_t step_idx0, uint64_t gid0,
                         const float rr[8], const float par_nom[4],
                         real *obs, real *reward, uint8_t *done, uint8_t *flags)
{
    int64_t t;
    for (t = 0; t < T; ++t)
        FN(qso_vec_step)(N, rec, par, actions + t * N * 4, kind, dt, integ, 1, randomise, seed,
                         step_idx0 + (uint64_t)t, gid0, rr, par_nom,
                         obs + t * N * 12, reward + t * N, done + t * N, flags + t * N, 0);
}

/* single drone + PID loop, run_sim_PID.py:43-54 (BASELINE config 1):
 * for each step: record state and u = PID(state_des, state); drone.step(u). */
void FN(qso_sim_pid)(int64_t T, real s[13], real u_prev[4], real sdes[13], const real par[4], real dt, int integ,
                     real *states_out, real *u_out)
{
    int64_t t; int i;
    for (t = 0; t < T; ++t) {
        real u[4];
        FN(qso_ctrl_pid)(sdes, s, par[0], u);
        if (states_out) for (i = 0; i < 13; ++i) states_out[13 * t + i] = s[i];
        if (u_out) for (i = 0; i < 4; ++i) u_out[4 * t + i] = u[i];
        FN(qso_drone_step)(s, u_prev, u, par, dt, integ);
    }
}

/* -------------------------------------------------------------------------
 * docking-v1 and hovering-v0
 * ---------------------------------------------------------------------- */

/* Vector driver for envs whose reset returns to a STORED per-env initial state:
 * docking-v1 = docking-v0 whose chaser start is jittered once at construction
 * (imitating_docking_env.py:34) and restored by every reset() (:193-204); also the
 * script-mutated env.chaser_ini_state case (run_expert_policy.py:44,63-64).
 * init [N][26] = chaser_ini_state, target_ini_state. */
void FN(qso_vec_step_stored_init)(int64_t N, real *rec, const real *par, const real *actions, int kind, real dt,
                                  int integ, int auto_reset, const real *init,
                                  real *obs, real *reward, uint8_t *done, uint8_t *flags, real *term_obs)
{
    int64_t i;
    for (i = 0; i < N; ++i) {
        real *r = rec + i * REC_LEN;
        real rew; int d, f, j;
        FN(qso_env_step)(r, actions + 4 * i, par + 4 * i, kind, dt, integ, obs + 12 * i, &rew, &d, &f);
        reward[i] = rew; done[i] = (uint8_t)d; flags[i] = (uint8_t)f;
        if (d && auto_reset) {
            if (term_obs) for (j = 0; j < 12; ++j) term_obs[12 * i + j] = obs[12 * i + j];
            FN(qso_env_reset)(r, init + 26 * i, init + 26 * i + 13, obs + 12 * i);
        }
    }
}

/* HoveringEnv.step, gym-docking/gym_docking/envs/hovering_env.py:47-78.
 * One drone; action in [0,1]^4 scaled by action_max = m g (:42,:51); obs = raw state;
 * reward :62-76; done :68.  state_des = (0,0,5, 0.., quat identity, 0..) (:31-35).
 * flags: bit0 = inside the +1 bonus ball (:63), bit3 = attitude limiter fired. */
void FN(qso_hover_step)(real s[13], real u_prev[4], const real a[4], const real par[4], real dt, int integ,
                        real *reward, int *done, int *flags)
{
    real mass = par[0];
    real lambda = (real)(K_KM / K_KF);
    real amax = (real)1.0 * mass * K_G;                       /* :42 */
    real f0 = amax * a[0], f1 = amax * a[1], f2 = amax * a[2], f3 = amax * a[3];
    real u[4];
    u[0] = f0 + f1 + f2 + f3;                                 /* rotor2control @ (.), :51 */
    u[1] = 0 * f0 + K_L * f1 + 0 * f2 + (-K_L) * f3;
    u[2] = (-K_L) * f0 + 0 * f1 + K_L * f2 + 0 * f3;
    u[3] = lambda * f0 + (-lambda) * f1 + lambda * f2 + (-lambda) * f3;
    int lim = FN(qso_drone_step)(s, u_prev, u, par, dt, integ);   /* :52 */
    real pe0 = 0 - s[0], pe1 = 0 - s[1], pe2 = (real)5.0 - s[2];  /* :57, state_des :31 */
    real ve0 = 0 - s[3], ve1 = 0 - s[4], ve2 = 0 - s[5];          /* :58 */
    const real qd[4] = {1, 0, 0, 0};
    real ed[3], en[3];
    FN(qso_quat2euler)(qd, ed);                               /* :59 */
    FN(qso_quat2euler)(s + 6, en);
    real ae0 = ed[0] - en[0], ae1 = ed[1] - en[1], ae2 = ed[2] - en[2];
    real we0 = 0 - s[10], we1 = 0 - s[11], we2 = 0 - s[12];   /* :60 */
    real npe = R_SQRT(pe0 * pe0 + pe1 * pe1 + pe2 * pe2), nve = R_SQRT(ve0 * ve0 + ve1 * ve1 + ve2 * ve2);
    real nae = R_SQRT(ae0 * ae0 + ae1 * ae1 + ae2 * ae2), nwe = R_SQRT(we0 * we0 + we1 * we1 + we2 * we2);
    int inside = (npe < (real)0.1) && (nve < (real)0.1);      /* :63 */
    real r_thre = inside ? (real)1.0 : (real)0.0;
    int d = (R_SQRT(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) > (real)100.0)
            || (R_SQRT(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]) > (real)100.0);   /* :68 */
    if (!d)
        *reward = r_thre + (real)0.1 - (real)0.01 * npe - (real)0.001 * nve - (real)0.01 * nae - (real)0.001 * nwe;  /* :71-74 */
    else
        *reward = (real)-0.1;                                 /* :76 */
    *done = d;
    *flags = (inside ? 1 : 0) | (d ? 2 : 0) | (lim ? 8 : 0);
}

/* vector hovering env with VecEnv auto-reset to the stored per-env ini_state
 * (hovering_env.py:80-82: reset() -> Drone.reset(ini_state): state restored, u zeroed).
 * st [N][17] = state 13 + last limited control 4; init [N][13]; obs [N][13] = state after step (or reset). */
void FN(qso_hover_vec_step)(int64_t N, real *st, const real *par, const real *actions, real dt, int integ,
                            int auto_reset, const real *init, real *obs, real *reward, uint8_t *done,
                            uint8_t *flags, real *term_obs)
{
    int64_t i; int j;
    for (i = 0; i < N; ++i) {
        real *s = st + 17 * i;
        real rew; int d, f;
        FN(qso_hover_step)(s, s + 13, actions + 4 * i, par + 4 * i, dt, integ, &rew, &d, &f);
        reward[i] = rew; done[i] = (uint8_t)d; flags[i] = (uint8_t)f;
        if (d && auto_reset) {
            if (term_obs) for (j = 0; j < 13; ++j) term_obs[13 * i + j] = s[j];
            for (j = 0; j < 13; ++j) s[j] = init[13 * i + j];
            for (j = 0; j < 4; ++j) s[13 + j] = 0;
        }
        for (j = 0; j < 13; ++j) obs[13 * i + j] = s[j];
    }
}

#define STREAM_INIT 3ull
/* construction-time jitter (stream INIT, ctr 0): kind 2 (docking-v1) chaser pos += U(-0.3,0.3)^3
 * (imitating_docking_env.py:34); kind 3 (hovering-v0) pos = (0,0,5) + U(-1,1)^3,
 * att = euler2quat(U(-0.2,0.2)^3) (hovering_env.py:23-24).  Same 16-bit lattice as qso_random_init.
 * out: init[26] (docking: chaser, target) or init[13] (hovering). */
void FN(qso_ctor_init)(uint64_t seed, uint64_t gid, int kind, float *init)
{
    uint32_t w[4];
    float u[8];
    int i;
    FN(qso_philox4x32_10)(seed, (STREAM_INIT << 48) | gid, 0, w);
    for (i = 0; i < 4; ++i) { u[2 * i] = u16f(w[i] & 0xFFFFu); u[2 * i + 1] = u16f(w[i] >> 16); }
    if (kind == 2) {
        for (i = 0; i < 26; ++i) init[i] = 0;
        init[0] = fmaf(fmaf(2.0f, u[0], -1.0f), 0.3f, 8.0f);
        init[1] = fmaf(fmaf(2.0f, u[1], -1.0f), 0.3f, -50.0f);
        init[2] = fmaf(fmaf(2.0f, u[2], -1.0f), 0.3f, 5.0f);
        init[6] = 1.0f;
        init[13] = 10.0f; init[14] = -50.0f; init[15] = 5.0f; init[19] = 1.0f;
    } else {
        float e0, e1, e2;
        for (i = 0; i < 13; ++i) init[i] = 0;
        init[0] = fmaf(2.0f, u[0], -1.0f);
        init[1] = fmaf(2.0f, u[1], -1.0f);
        init[2] = fmaf(fmaf(2.0f, u[2], -1.0f), 1.0f, 5.0f);
        e0 = fmaf(2.0f, u[3], -1.0f) * 0.2f; e1 = fmaf(2.0f, u[4], -1.0f) * 0.2f; e2 = fmaf(2.0f, u[5], -1.0f) * 0.2f;
        {
            float cy = cosf(e2 * 0.5f), sy = sinf(e2 * 0.5f), cp = cosf(e1 * 0.5f), sp = sinf(e1 * 0.5f);
            float cr = cosf(e0 * 0.5f), sr = sinf(e0 * 0.5f);
            init[6] = cr * cp * cy - sr * sp * sy;
            init[7] = sr * cp * cy - cr * sp * sy;
            init[8] = sr * cp * sy + cr * sp * cy;
            init[9] = cr * cp * sy + sr * sp * cy;
        }
    }
}

/* -------------------------------------------------------------------------
 * GAE(lambda) of the PPO2 Runner, rl_baselines/ppo2/ppo2.py:507-520.
 * float32 rewards / values in, float32 advantages / returns out; the recurrence itself runs in
 * double in the reference (`1.0 - bool` is float64, and the chained assignment
 * `mb_advs[step] = last_gae_lam = ...` keeps the un-rounded value); only `self.gamma * nextvalues` is a
 * Python float times a float32 array, i.e. a float32 product -- restated as such.
 * dones[t] is the done flag BEFORE step t (mb_dones, :474); last_dones = self.dones after the last step.
 * Compiled in both builds with identical float/double types (it does not depend on `real`).
 * ---------------------------------------------------------------------- */
void FN(qso_gae)(int64_t T, int64_t N, const float *rewards, const float *values, const uint8_t *dones,
                 const float *last_values, const uint8_t *last_dones, double gamma, double lam,
                 float *advs, float *returns)
{
    int64_t i, t;
    for (i = 0; i < N; ++i) {
        double last = 0.0;
        for (t = T - 1; t >= 0; --t) {
            double nonterm, nextv;
            if (t == T - 1) { nonterm = 1.0 - (double)(last_dones[i] != 0); nextv = (double)last_values[i]; }   /* :512-514 */
            else { nonterm = 1.0 - (double)(dones[(t + 1) * N + i] != 0); nextv = (double)values[(t + 1) * N + i]; } /* :516-517 */
            double gv = (double)((float)gamma * (float)nextv);                /* float32 product */
            double delta = (double)rewards[t * N + i] + gv * nonterm - (double)values[t * N + i];                  /* :518 */
            last = delta + gamma * lam * nonterm * last;                                                          /* :519 */
            advs[t * N + i] = (float)last;
            returns[t * N + i] = advs[t * N + i] + values[t * N + i];                                              /* :520 (float32 + float32) */
        }
    }
}

/* -------------------------------------------------------------------------
 * the PID expert of run_expert_policy.py:49-69 / run_expert_record.py:121-136.
 * des_vel = kp (p_target + (-0.2,0,0) - p_chaser) + kd (-v_chaser)                      (:57)
 * state_des[3:6] = des_vel except on the first step of an episode                       (:58-59)
 * u = vel_controller(state_des, state_chaser, state_last)  (state_last aliases the current state: e_dv = 0)  (:60)
 * action = (inv(rotor2control) @ u - action_mean) / action_std, NOT clipped               (:63)
 * sdes [13] is the expert's persistent state_des (pos = the chaser's initial position; [6:12] rewritten by the
 * controller).  `first` = first step of the episode (idx_per_epi == 0).
 * ---------------------------------------------------------------------- */
void FN(qso_expert_action)(real sdes[13], const real sc[13], const real st[13], int first, real kp, real kd,
                           real mass, real action[4], real u_out[4])
{
    real u[4];
    int i;
    if (!first) {
        sdes[3] = kp * (st[0] + (real)-0.2 - sc[0]) + kd * (-sc[3]);
        sdes[4] = kp * (st[1] + 0 - sc[1]) + kd * (-sc[4]);
        sdes[5] = kp * (st[2] + 0 - sc[2]) + kd * (-sc[5]);
    }
    FN(qso_ctrl_vel)(sdes, sc, sc, mass, u);
    /* inverse of rotor2control (quadrotor.py:56-59) */
    real lambda = (real)(K_KM / K_KF);
    real f4 = u[0] / 4, a = (real)1.0 / (2 * K_L), b = (real)1.0 / (4 * lambda);
    real f[4];
    f[0] = f4 - a * u[2] + b * u[3];
    f[1] = f4 + a * u[1] - b * u[3];
    f[2] = f4 + a * u[2] + b * u[3];
    f[3] = f4 - a * u[1] - b * u[3];
    real mean = mass * K_G / (real)2.0;
    for (i = 0; i < 4; ++i) action[i] = (f[i] - mean) / mean;
    if (u_out) for (i = 0; i < 4; ++i) u_out[i] = u[i];
}

int FN(qso_real_size)(void) { return (int)sizeof(real); }
