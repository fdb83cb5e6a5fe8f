/*
 * caar_oracle.c — CPU oracle for compute_and_apply_rhs.  TEST INFRASTRUCTURE ONLY
 * (see caar_oracle.h: never linked or called by the product path).
 *
 * Restates, in plain C with run-time dimensions, the algorithm of
 *   P: compute_and_apply_rhs_test/cxx/pointers_only/compute_and_apply_rhs.cpp
 *   S: compute_and_apply_rhs_test/cxx/pointers_only/sphere_operators.cpp
 *   F: compute_and_apply_rhs_test/fortran/routine_mod.F90 (+ derivative_mod_base.F90)
 * Every floating-point expression keeps P/S's operand order and association so
 * that, built with -ffp-contract=off, the results are bit-identical to the
 * reference C++ built by g++ -O3 for x86-64 (checked in tests/test_oracle.py).
 *
 * Parity status: PINNED (see caar_oracle.h).
 */
#include "caar_oracle.h"

#include <math.h>
#include <stdlib.h>

/* ---- index helpers (row-major, last index fastest; cf. test_macros.hpp:1-54) */
#define I2(i, j, n2) ((size_t)(i) * (n2) + (j))
#define I3(i, j, k, n2, n3) (((size_t)(i) * (n2) + (j)) * (n3) + (k))
#define I4(i, j, k, l, n2, n3, n4) ((((size_t)(i) * (n2) + (j)) * (n3) + (k)) * (n4) + (l))

/* S:9-48.  s[np][np] -> ds[np][np][2].  Dinv[np][np][2][2]. */
void oracle_gradient_sphere(int np, const double *s, const double *Dvv,
                            const double *Dinv, double rrearth, double *ds) {
  double v1[8 * 8], v2[8 * 8];
  for (int j = 0; j < np; ++j) {
    for (int l = 0; l < np; ++l) {
      double dsdx = 0, dsdy = 0;
      for (int i = 0; i < np; ++i) {
        dsdx += Dvv[I2(i, l, np)] * s[I2(i, j, np)]; /* S:30 */
        dsdy += Dvv[I2(i, l, np)] * s[I2(j, i, np)]; /* S:31 */
      }
      v1[I2(l, j, np)] = dsdx * rrearth; /* S:34 */
      v2[I2(j, l, np)] = dsdy * rrearth; /* S:35 */
    }
  }
  for (int j = 0; j < np; ++j) {
    for (int i = 0; i < np; ++i) {
      const double *Di = &Dinv[I4(i, j, 0, 0, np, 2, 2)];
      /* S:43-47: ds_c = Dinv[0][c]*v1 + Dinv[1][c]*v2 */
      ds[I3(i, j, 0, np, 2)] = Di[0] * v1[I2(i, j, np)] + Di[2] * v2[I2(i, j, np)];
      ds[I3(i, j, 1, np, 2)] = Di[1] * v1[I2(i, j, np)] + Di[3] * v2[I2(i, j, np)];
    }
  }
}

/* S:50-89.  v[np][np][2] -> div[np][np]. */
void oracle_divergence_sphere(int np, const double *v, const double *Dvv,
                              const double *Dinv, const double *metdet,
                              const double *rmetdet, double rrearth, double *div) {
  double gv[8 * 8 * 2];
  for (int i = 0; i < np; ++i) {
    for (int j = 0; j < np; ++j) {
      const double *Di = &Dinv[I4(i, j, 0, 0, np, 2, 2)];
      double u0 = v[I3(i, j, 0, np, 2)], u1 = v[I3(i, j, 1, np, 2)];
      gv[I3(i, j, 0, np, 2)] = metdet[I2(i, j, np)] * (Di[0] * u0 + Di[1] * u1); /* S:66-67 */
      gv[I3(i, j, 1, np, 2)] = metdet[I2(i, j, np)] * (Di[2] * u0 + Di[3] * u1); /* S:68-69 */
    }
  }
  for (int i = 0; i < np; ++i) {
    for (int j = 0; j < np; ++j) {
      double dudx = 0., dvdy = 0.;
      for (int k = 0; k < np; ++k) {
        dudx += Dvv[I2(k, i, np)] * gv[I3(k, j, 0, np, 2)]; /* S:81 */
        dvdy += Dvv[I2(k, j, np)] * gv[I3(i, k, 1, np, 2)]; /* S:82 */
      }
      div[I2(i, j, np)] = (dudx + dvdy) * rmetdet[I2(i, j, np)] * rrearth; /* S:85 */
    }
  }
}

/* S:91-129.  v[np][np][2] -> vort[np][np].  D[np][np][2][2]. */
void oracle_vorticity_sphere(int np, const double *v, const double *Dvv,
                             const double *D, const double *rmetdet,
                             double rrearth, double *vort) {
  double vcov[8 * 8 * 2];
  for (int i = 0; i < np; ++i) {
    for (int j = 0; j < np; ++j) {
      const double *Dm = &D[I4(i, j, 0, 0, np, 2, 2)];
      double u0 = v[I3(i, j, 0, np, 2)], u1 = v[I3(i, j, 1, np, 2)];
      vcov[I3(i, j, 0, np, 2)] = Dm[0] * u0 + Dm[2] * u1; /* S:106-107 */
      vcov[I3(i, j, 1, np, 2)] = Dm[1] * u0 + Dm[3] * u1; /* S:108-109 */
    }
  }
  for (int i = 0; i < np; ++i) {
    for (int j = 0; j < np; ++j) {
      double dudy = 0., dvdx = 0.;
      for (int k = 0; k < np; ++k) {
        dvdx += Dvv[I2(k, i, np)] * vcov[I3(k, j, 1, np, 2)]; /* S:121 */
        dudy += Dvv[I2(k, j, np)] * vcov[I3(i, k, 0, np, 2)]; /* S:122 */
      }
      vort[I2(i, j, np)] = (dvdx - dudy) * rmetdet[I2(i, j, np)] * rrearth; /* S:125 */
    }
  }
}

/* P:280-312.  Bottom-up hydrostatic integral; phii is the interface value. */
void oracle_preq_hydrostatic(int np, int nlev, const double *phis,
                             const double *T_v, const double *p,
                             const double *dp, double Rgas, double *phi) {
  const int npp = np * np;
  for (int q = 0; q < npp; ++q) { /* columns are independent (P:287-311) */
    double hkk, hkl, phii;
    int k = nlev - 1;
    hkk = 0.5 * dp[(size_t)k * npp + q] / p[(size_t)k * npp + q]; /* P:291 */
    hkl = 2.0 * hkk;
    phii = Rgas * T_v[(size_t)k * npp + q] * hkl;                      /* P:293 */
    phi[(size_t)k * npp + q] = phis[q] + Rgas * T_v[(size_t)k * npp + q] * hkk; /* P:294 */
    for (k = nlev - 2; k > 0; --k) {
      hkk = 0.5 * dp[(size_t)k * npp + q] / p[(size_t)k * npp + q]; /* P:300 */
      hkl = 2.0 * hkk;
      phi[(size_t)k * npp + q] = phis[q] + phii + Rgas * T_v[(size_t)k * npp + q] * hkk; /* P:303 */
      phii = phii + Rgas * T_v[(size_t)k * npp + q] * hkl;                                /* P:302 */
    }
    hkk = 0.5 * dp[q] / p[q];                            /* P:308 */
    phi[q] = phis[q] + phii + Rgas * T_v[q] * hkk;       /* P:309 */
  }
}

/* P:314-352.  Top-down omega/p integral; suml = sum of divdp above. */
void oracle_preq_omega_ps(int np, int nlev, const double *p,
                          const double *vgrad_p, const double *divdp,
                          double *omega_p) {
  const int npp = np * np;
  for (int q = 0; q < npp; ++q) {
    double ckk, ckl, term, suml;
    ckk = 0.5 / p[q];                                  /* P:323 */
    term = divdp[q];
    omega_p[q] = vgrad_p[q] / p[q] - ckk * term;       /* P:325 */
    suml = term;
    for (int k = 1; k < nlev - 1; ++k) {
      size_t o = (size_t)k * npp + q;
      ckk = 0.5 / p[o];                                /* P:333 */
      ckl = 2.0 * ckk;
      term = divdp[o];
      omega_p[o] = vgrad_p[o] / p[o] - ckl * suml - ckk * term; /* P:336-337 */
      suml += term;
    }
    {
      size_t o = (size_t)(nlev - 1) * npp + q;
      ckk = 0.5 / p[o];                                /* P:345 */
      ckl = 2.0 * ckk;
      term = divdp[o];
      omega_p[o] = vgrad_p[o] / p[o] - ckl * suml - ckk * term; /* P:348-349 */
    }
  }
}

/* rsplit == 0 (Eulerian vertical coordinate): the vertical mass flux at the interfaces and the
 * vertical advection of T and v.  PARITY UNPINNED: the reference states this branch only in
 * files it never builds — X = compute_and_apply_rhs_test/fortran/routine_extracted.F90:224-262
 * (eta_dot_dpdn) and cxx/level_vectorized_ppscan/CaarFunctor.hpp:505-547 (preq_vertadv, "UNTESTED",
 * "Not currently used"; HOMME's prim_si_mod::preq_vertadv, eq. CCM2 (3.b.1)) — and holds no
 * output for it.  eta_dot: [nlev+1][np][np]; v, v_vadv: [nlev][np][np][2]. */
static void oracle_eulerian_vertical(int np, int nlev, const double *divdp, const double *hybi,
                                     const double *dp, const double *T, const double *v,
                                     double *eta_dot, double *rdp, double *T_vadv, double *v_vadv) {
  const int npp = np * np;
  for (int q = 0; q < npp; ++q) {
    double sdot_sum = 0.0;                                /* X:221 */
    for (int k = 0; k < nlev; ++k) {
      sdot_sum = sdot_sum + divdp[k * npp + q];           /* X:237 */
      eta_dot[(k + 1) * npp + q] = sdot_sum;              /* X:238 */
    }
    for (int k = 0; k < nlev - 1; ++k)                    /* X:249-251 */
      eta_dot[(k + 1) * npp + q] = hybi[k + 1] * sdot_sum - eta_dot[(k + 1) * npp + q];
    eta_dot[q] = 0.0;                                     /* X:253 */
    eta_dot[nlev * npp + q] = 0.0;                        /* X:254 */
  }
  for (int o = 0; o < nlev * npp; ++o) rdp[o] = 1.0 / dp[o]; /* X:118 */
  /* preq_vertadv, CaarFunctor.hpp:505-547 */
  for (int k = 0; k < nlev; ++k) {
    for (int q = 0; q < npp; ++q) {
      const int o = k * npp + q;
      double tv = 0.0, vv0 = 0.0, vv1 = 0.0;
      if (k == 0) {                                       /* CaarFunctor.hpp:513-522 */
        const double facp = 0.5 * rdp[o] * eta_dot[o + npp];
        tv = facp * (T[o + npp] - T[o]);
        vv0 = facp * (v[2 * (o + npp)] - v[2 * o]);
        vv1 = facp * (v[2 * (o + npp) + 1] - v[2 * o + 1]);
      } else if (k < nlev - 1) {                          /* CaarFunctor.hpp:524-537 */
        const double facp = 0.5 * rdp[o] * eta_dot[o + npp];
        const double facm = 0.5 * rdp[o] * eta_dot[o];
        tv = facp * (T[o + npp] - T[o]) + facm * (T[o] - T[o - npp]);
        vv0 = facp * (v[2 * (o + npp)] - v[2 * o]) + facm * (v[2 * o] - v[2 * (o - npp)]);
        vv1 = facp * (v[2 * (o + npp) + 1] - v[2 * o + 1]) + facm * (v[2 * o + 1] - v[2 * (o - npp) + 1]);
      } else {                                            /* CaarFunctor.hpp:538-546 */
        const double facm = 0.5 * rdp[o] * eta_dot[o];
        tv = facm * (T[o] - T[o - npp]);
        vv0 = facm * (v[2 * o] - v[2 * (o - npp)]);
        vv1 = facm * (v[2 * o + 1] - v[2 * (o - npp) + 1]);
      }
      T_vadv[o] = tv;
      v_vadv[2 * o] = vv0;
      v_vadv[2 * o + 1] = vv1;
    }
  }
}

int oracle_compute_and_apply_rhs_repeat(const oracle_arrays *a, const oracle_params *c, int reps) {
  int rc = 0;
  for (int i = 0; i < reps && rc == 0; ++i) rc = oracle_compute_and_apply_rhs(a, c);
  return rc;
}

/* P:15-278. */
int oracle_compute_and_apply_rhs(const oracle_arrays *a, const oracle_params *c) {
  const int np = c->np, nlev = c->nlev, tl = c->timelevels, qd = c->qsize_d;
  const int npp = np * np;
  const size_t blk = (size_t)nlev * npp; /* one scalar field block */
  if (np > 8) return -2;

  /* element-private temporaries (P:18-35): 13 field blocks + per-level scratch */
  const int eulerian = c->rsplit == 0; /* routine_extracted.F90:227: "if (rsplit>0) ... else" */
  if (eulerian && !c->hybi) return -3;
  double *buf = (double *)calloc(blk * 18 + (size_t)npp * 5, sizeof(double));
  if (!buf) return -1;
  double *T_v = buf;
  double *divdp = T_v + blk;
  double *grad_p = divdp + blk; /* [nlev][np][np][2] */
  double *omega_p_tmp = grad_p + 2 * blk;
  double *p = omega_p_tmp + blk;
  double *ttens = p + blk;
  double *vdp = ttens + blk; /* [nlev][np][np][2] */
  double *vgrad_p = vdp + 2 * blk;
  double *vort = vgrad_p + blk;
  double *vt1 = vort + blk; /* vtens1 */
  double *vt2 = vt1 + blk;  /* vtens2 */
  double *Ephi = vt2 + blk; /* [np][np] */
  double *vgrad_T = Ephi + npp;
  double *vtemp = vgrad_T + npp; /* [np][np][2] */
  /* rsplit == 0 only (zero otherwise, routine_extracted.F90:229-231) */
  double *eta_dot = vtemp + 2 * npp; /* [nlev+1][np][np] */
  double *T_vadv = eta_dot + blk + npp;
  double *v_vadv = T_vadv + blk; /* [nlev][np][np][2] */
  double *rdp = v_vadv + 2 * blk;

  for (int ie = c->nets; ie < c->nete; ++ie) {
    const double *Dinv = a->elem_Dinv + (size_t)ie * npp * 4;
    const double *Dm = a->elem_D + (size_t)ie * npp * 4;
    const double *metdet = a->elem_metdet + (size_t)ie * npp;
    const double *rmetdet = a->elem_rmetdet + (size_t)ie * npp;
    const double *fcor = a->elem_fcor + (size_t)ie * npp;
    const double *spheremp = a->elem_spheremp + (size_t)ie * npp;
    const double *phis = a->elem_state_phis + (size_t)ie * npp;

    const double *dp_n0 = a->elem_state_dp3d + ((size_t)ie * tl + c->n0) * blk;
    const double *v_n0 = a->elem_state_v + ((size_t)ie * tl + c->n0) * blk * 2;
    const double *T_n0 = a->elem_state_T + ((size_t)ie * tl + c->n0) * blk;
    double *vn0 = a->elem_derived_vn0 + (size_t)ie * blk * 2;

    /* S1 pressure at mid-levels (P:78-97) */
    for (int q = 0; q < npp; ++q) p[q] = c->hyai0 * c->ps0 + 0.5 * dp_n0[q]; /* P:84 */
    for (int k = 1; k < nlev; ++k)
      for (int q = 0; q < npp; ++q)
        p[k * npp + q] = p[(k - 1) * npp + q] + 0.5 * dp_n0[(k - 1) * npp + q] +
                         0.5 * dp_n0[k * npp + q]; /* P:94-96 */

    /* S2 (P:101-124) */
    for (int k = 0; k < nlev; ++k) {
      oracle_gradient_sphere(np, p + k * npp, c->Dvv, Dinv, c->rrearth, grad_p + (size_t)k * npp * 2);
      for (int q = 0; q < npp; ++q) {
        size_t o = (size_t)k * npp + q;
        double v1 = v_n0[2 * o], v2 = v_n0[2 * o + 1];
        vgrad_p[o] = v1 * grad_p[2 * o] + v2 * grad_p[2 * o + 1]; /* P:111-112 */
        vdp[2 * o] = v1 * dp_n0[o];                               /* P:114 */
        vdp[2 * o + 1] = v2 * dp_n0[o];                           /* P:115 */
        vn0[2 * o] += c->eta_ave_w * vdp[2 * o];                  /* P:117 */
        vn0[2 * o + 1] += c->eta_ave_w * vdp[2 * o + 1];          /* P:118 */
      }
      oracle_divergence_sphere(np, vdp + (size_t)k * npp * 2, c->Dvv, Dinv, metdet, rmetdet,
                               c->rrearth, divdp + k * npp);
      oracle_vorticity_sphere(np, v_n0 + (size_t)k * npp * 2, c->Dvv, Dm, rmetdet, c->rrearth,
                              vort + k * npp);
    }

    /* S3 virtual temperature (P:126-156) */
    if (c->qn0 == -1) {
      for (size_t o = 0; o < blk; ++o) T_v[o] = T_n0[o]; /* P:135 */
    } else {
      const double *Qdp = a->elem_state_Qdp + (((size_t)ie * qd + 0) * 2 + c->qn0) * blk; /* P:143 */
      for (size_t o = 0; o < blk; ++o) {
        double Qt = Qdp[o] / dp_n0[o];                                            /* P:150 */
        T_v[o] = T_n0[o] * (1.0 + (c->Rwater_vapor / c->Rgas - 1.0) * Qt);        /* P:151 */
      }
    }

    double *phi = a->elem_derived_phi + (size_t)ie * blk;
    oracle_preq_hydrostatic(np, nlev, phis, T_v, p, dp_n0, c->Rgas, phi); /* P:161 */
    oracle_preq_omega_ps(np, nlev, p, vgrad_p, divdp, omega_p_tmp);       /* P:162 */

    if (eulerian) oracle_eulerian_vertical(np, nlev, divdp, c->hybi, dp_n0, T_n0, v_n0, eta_dot, rdp, T_vadv, v_vadv);

    /* S6 accumulators (P:164-183); eta_dot_dpdn_tmp == 0 when vertically Lagrangian */
    double *omega_p = a->elem_derived_omega_p + (size_t)ie * blk;
    double *eta = a->elem_derived_eta_dot_dpdn + (size_t)ie * (blk + npp);
    for (size_t o = 0; o < blk; ++o) {
      eta[o] += c->eta_ave_w * (eulerian ? eta_dot[o] : 0.0); /* P:172, X:271-272 */
      omega_p[o] += c->eta_ave_w * omega_p_tmp[o];             /* P:173 */
    }
    for (int q = 0; q < npp; ++q) eta[blk + q] += c->eta_ave_w * (eulerian ? eta_dot[blk + q] : 0.0); /* P:181, X:276-277 */

    /* S7 tendencies (P:185-233) */
    const double *pecnd = a->elem_derived_pecnd + (size_t)ie * blk;
    for (int k = 0; k < nlev; ++k) {
      for (int q = 0; q < npp; ++q) {
        size_t o = (size_t)k * npp + q;
        double v1 = v_n0[2 * o], v2 = v_n0[2 * o + 1];
        Ephi[q] = 0.5 * (v1 * v1 + v2 * v2) + phi[o] + pecnd[o]; /* P:196 */
      }
      oracle_gradient_sphere(np, T_n0 + k * npp, c->Dvv, Dinv, c->rrearth, vtemp); /* P:200 */
      for (int q = 0; q < npp; ++q) {
        size_t o = (size_t)k * npp + q;
        double v1 = v_n0[2 * o], v2 = v_n0[2 * o + 1];
        vgrad_T[q] = v1 * vtemp[2 * q] + v2 * vtemp[2 * q + 1]; /* P:209 */
      }
      oracle_gradient_sphere(np, Ephi, c->Dvv, Dinv, c->rrearth, vtemp); /* P:213 */
      for (int q = 0; q < npp; ++q) {
        size_t o = (size_t)k * npp + q;
        double gpterm = T_v[o] / p[o];                          /* P:219 */
        double glnps1 = c->Rgas * gpterm * grad_p[2 * o];       /* P:221 */
        double glnps2 = c->Rgas * gpterm * grad_p[2 * o + 1];   /* P:222 */
        double v1 = v_n0[2 * o], v2 = v_n0[2 * o + 1];
        /* v_vadv == T_vadv == 0 (P:27-28) unless rsplit == 0, kept in the expression as in
         * P:227-231 / X:326-338 */
        vt1[o] = -v_vadv[2 * o] + v2 * (fcor[q] + vort[o]) - vtemp[2 * q] - glnps1;         /* P:227 */
        vt2[o] = -v_vadv[2 * o + 1] - v1 * (fcor[q] + vort[o]) - vtemp[2 * q + 1] - glnps2; /* P:228 */
        ttens[o] = -T_vadv[o] - vgrad_T[q] + c->kappa * T_v[o] * omega_p_tmp[o];             /* P:230-231 */
      }
    }

    /* S8 update (P:236-257) */
    double *v_np1 = a->elem_state_v + ((size_t)ie * tl + c->np1) * blk * 2;
    double *T_np1 = a->elem_state_T + ((size_t)ie * tl + c->np1) * blk;
    double *dp_np1 = a->elem_state_dp3d + ((size_t)ie * tl + c->np1) * blk;
    const double *v_nm1 = a->elem_state_v + ((size_t)ie * tl + c->nm1) * blk * 2;
    const double *T_nm1 = a->elem_state_T + ((size_t)ie * tl + c->nm1) * blk;
    const double *dp_nm1 = a->elem_state_dp3d + ((size_t)ie * tl + c->nm1) * blk;
    for (int k = 0; k < nlev; ++k) {
      for (int q = 0; q < npp; ++q) {
        size_t o = (size_t)k * npp + q;
        v_np1[2 * o] = spheremp[q] * (v_nm1[2 * o] + c->dt2 * vt1[o]);         /* P:251 */
        v_np1[2 * o + 1] = spheremp[q] * (v_nm1[2 * o + 1] + c->dt2 * vt2[o]); /* P:252 */
        T_np1[o] = spheremp[q] * (T_nm1[o] + c->dt2 * ttens[o]);               /* P:253 */
        if (eulerian) /* X:515-517 */
          dp_np1[o] = spheremp[q] * (dp_nm1[o] - c->dt2 * (divdp[o] + eta_dot[o + npp] - eta_dot[o]));
        else
          dp_np1[o] = spheremp[q] * (dp_nm1[o] - c->dt2 * divdp[o]);           /* P:254 */
      }
    }
  }
  free(buf);
  return 0;
}

/* P:353-370 */
double oracle_compute_norm(const double *field, long length) {
  double norm = 0, c = 0; /* Kahan summation: never build this file with -ffast-math */
  for (long i = 0; i < length; ++i) {
    double y = field[i] * field[i] - c;
    double temp = norm + y;
    c = (temp - norm) - y;
    norm = temp;
  }
  return sqrt(norm);
}

/* P:372-399 */
void oracle_state_norms(const oracle_arrays *a, const oracle_params *c, double out[3]) {
  const size_t blk = (size_t)c->nlev * c->np * c->np;
  double vn = 0, tn = 0, dn = 0;
  for (int ie = c->nets; ie < c->nete; ++ie) {
    const double *v = a->elem_state_v + ((size_t)ie * c->timelevels + c->np1) * blk * 2;
    const double *T = a->elem_state_T + ((size_t)ie * c->timelevels + c->np1) * blk;
    const double *dp = a->elem_state_dp3d + ((size_t)ie * c->timelevels + c->np1) * blk;
    vn += pow(oracle_compute_norm(v, (long)blk * 2), 2);
    tn += pow(oracle_compute_norm(T, (long)blk), 2);
    dn += pow(oracle_compute_norm(dp, (long)blk), 2);
  }
  out[0] = sqrt(vn);
  out[1] = sqrt(tn);
  out[2] = sqrt(dn);
}

/* data_structures.cpp:42-92 */
void oracle_init_arrays(const oracle_arrays *a, int np, int nlev, int qd, int tl, int num_elems) {
  const int npp = np * np;
  const size_t blk = (size_t)nlev * npp;
  for (int ie = 0; ie < num_elems; ++ie) {
    double *eta = a->elem_derived_eta_dot_dpdn + (size_t)ie * (blk + npp);
    for (size_t o = 0; o < blk + npp; ++o) eta[o] = 0.0;
    double *Qdp = a->elem_state_Qdp + (size_t)ie * qd * 2 * blk;
    for (size_t o = 0; o < (size_t)qd * 2 * blk; ++o) Qdp[o] = 0.0;
    for (int ip = 0; ip < np; ++ip) {
      for (int jp = 0; jp < np; ++jp) {
        const double iie = ie + 1, iip = ip + 1, jjp = jp + 1;
        const size_t q = (size_t)ip * np + jp, eq = (size_t)ie * npp + q;
        a->elem_fcor[eq] = sin(iip + jjp);
        a->elem_metdet[eq] = iip * jjp;
        a->elem_rmetdet[eq] = 1. / a->elem_metdet[eq];
        a->elem_spheremp[eq] = 2 * iip;
        a->elem_state_phis[eq] = iip + jjp;
        double *Dm = a->elem_D + eq * 4, *Di = a->elem_Dinv + eq * 4;
        Dm[0] = 1.0; Dm[1] = 0.0; Dm[2] = 0.0; Dm[3] = 2.0;
        Di[0] = 1.0; Di[1] = 0.0; Di[2] = 0.0; Di[3] = 0.5;
        for (int il = 0; il < nlev; ++il) {
          const double iil = il + 1;
          const size_t o = (size_t)ie * blk + (size_t)il * npp + q;
          a->elem_derived_phi[o] = cos(iip + 3 * jjp) + iil;
          a->elem_derived_vn0[2 * o] = 1.0;
          a->elem_derived_vn0[2 * o + 1] = 1.0;
          a->elem_derived_pecnd[o] = 1.0;
          a->elem_derived_omega_p[o] = jjp * jjp;
          Qdp[(size_t)il * npp + q] = 1.0 + sin(iip * jjp * iil); /* [q=0][t=0] */
          for (int it = 0; it < tl; ++it) {
            const double iit = it + 1;
            const size_t s = ((size_t)ie * tl + it) * blk + (size_t)il * npp + q;
            a->elem_state_dp3d[s] = 10.0 * iil + iie + iip + jjp + iit;
            a->elem_state_v[2 * s] = 1.0 + 0.5 * iil + iip + jjp + 0.2 * iie + 2.0 * iit;
            a->elem_state_v[2 * s + 1] = 1.0 + 0.5 * iil + iip + jjp + 0.2 * iie + 3.0 * iit;
            a->elem_state_T[s] = 1000.0 - iil - iip - jjp + 0.1 * iie + iit;
          }
        }
      }
    }
  }
}

/* data_structures.cpp:152-162; f32 rounding per main.F90:83-96 */
void oracle_init_dvv_np4(double *Dvv, int f32_rounded) {
  static const double values[16] = {
      -3.0000000000000000, -0.80901699437494745, 0.30901699437494745, -0.50000000000000000,
      4.0450849718747373,  0.00000000000000000,  -1.11803398874989490, 1.54508497187473700,
      -1.5450849718747370, 1.11803398874989490,  0.00000000000000000,  -4.04508497187473730,
      0.5000000000000000,  -0.30901699437494745, 0.80901699437494745,  3.000000000000000000};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double x = values[j * 4 + i];
      Dvv[i * 4 + j] = f32_rounded ? (double)(float)x : x;
    }
}

/* Legendre P_n(x) by the three-term recurrence. */
static double legendre(int n, double x) {
  double p0 = 1.0, p1 = x;
  if (n == 0) return p0;
  for (int k = 2; k <= n; ++k) {
    double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
    p0 = p1;
    p1 = pk;
  }
  return p1;
}

/* GLL nodes = roots of (1-x^2) P'_N(x), N = np-1; Newton on q(x) = P_{N+1} - P_{N-1}. */
void oracle_init_dvv_gll(int np, double *Dvv) {
  const int N = np - 1;
  double x[16];
  const double pi = 3.14159265358979323846;
  x[0] = -1.0;
  x[N] = 1.0;
  for (int i = 1; i < N; ++i) {
    double xi = -cos(pi * i / N);
    for (int it = 0; it < 100; ++it) {
      /* f = P'_N up to a factor: (1-x^2)P'_N = N (P_{N-1} - x P_N) */
      double pn = legendre(N, xi), pnm1 = legendre(N - 1, xi);
      double f = N * (pnm1 - xi * pn);                  /* (1-x^2) P'_N */
      double fp = -(double)N * (N + 1) * pn;            /* d/dx[(1-x^2)P'_N] = -N(N+1)P_N */
      double dx = f / fp;
      xi -= dx;
      if (fabs(dx) < 1e-16) break;
    }
    x[i] = xi;
  }
  for (int i = 0; i <= N; ++i)
    for (int j = 0; j <= N; ++j) {
      double d;
      if (i != j)
        d = legendre(N, x[i]) / (legendre(N, x[j]) * (x[i] - x[j]));
      else if (i == 0)
        d = -0.25 * N * (N + 1);
      else if (i == N)
        d = 0.25 * N * (N + 1);
      else
        d = 0.0;
      Dvv[i * np + j] = d;
    }
}
