/*
 * sphere_ops_oracle.c — CPU oracle for the sphere operators NEXT TO the CAAR path.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as caar_oracle.c: only tests/, smoke() and bench.py's
 * cpu_baseline leg may call it; the product never does).
 *
 * What is restated: the multi-level operators of
 *     compute_and_apply_rhs_test/cxx/level_vectorized_ppscan/SphereOperators.hpp   (cited as K:)
 *   gradient_sphere K:229-269, gradient_sphere_update K:271-312, divergence_sphere K:315-358,
 *   divergence_sphere_update K:363-403, vorticity_sphere_vector K:452-490, divergence_sphere_wk
 *   K:494-534, laplace_simple K:538-550, laplace_tensor K:556-596, laplace_tensor_replace K:600-637,
 *   curl_sphere_wk_testcov K:640-690,
 *   grad_sphere_wk_testcov K:694-770, vlaplace_sphere_wk_cartesian_reduced K:849-915,
 *   vlaplace_sphere_wk_contra K:938-993.
 *
 * PARITY UNPINNED: these functions exist in the reference only as Kokkos device functions of variants it
 * cannot build here (Kokkos absent), nothing in the reference calls or tests them (the Euler-step functor
 * that would is inconsistent, EulerStepFunctor.hpp:66-67 vs K:362-370), its element data structure holds
 * no mp / metinv / tensorVisc / vec_sph2cart at all (Elements.hpp:19-26), and no output of them exists in
 * the reference tree.  What pins them instead (tests/test_sphere_ops.py):
 *   (1) the INDEX MAPPING used here is validated on the three operators that ARE pinned: the K: forms of
 *       gradient/divergence/vorticity restated through the very same accessor macros reproduce the pinned
 *       pointers_only operators (caar_oracle.c, bit-identical to the reference's C++) to rounding;
 *   (2) defining identities: divergence_sphere_wk is the negative adjoint of the pinned gradient_sphere
 *       under the spheremp-weighted inner product; laplace_simple = divergence_sphere_wk o gradient_sphere;
 *       laplace_tensor with the identity tensor = laplace_simple; vlaplace forms against their compositions;
 *   (3) an independent numpy restatement from HOMME's published Fortran formulas (einsum form).
 *
 * Index conventions.  K: uses field(igp, jgp) == Fortran (j, i) and tensor(p, q, igp, jgp) == Fortran
 * (i, j, q+1, p+1), dvv(x, y) == Fortran Dvv(y+1, x+1) (Elements.cpp:176-199, Derivative.cpp:11-23).
 * This repository's arrays use the pointers_only convention: field[a][b] == Fortran
 * (a+1, b+1), tensor[a][b][r][c] == Fortran (a+1, b+1, r+1, c+1), Dvv[i][j] == Fortran Dvv(i+1, j+1).
 * The accessor macros below take K:'s indices and address this repository's layout, so each formula can
 * be read side by side with its K: lines:
 *     K: f(igp, jgp)              ->  f[jgp][igp]
 *     K: v(h, igp, jgp)           ->  v[jgp][igp][h]
 *     K: t(p, q, igp, jgp)        ->  t[jgp][igp][q][p]
 *     K: vec_sph2cart(p, k, i, j) ->  s2c[j][i][k][p]      ([np][np][3][2])
 *     K: dvv(x, y)                ->  Dvv[y][x]
 */
#include <stddef.h>

#include "caar_oracle.h"

#define MAXNP 8
#define KF(f, igp, jgp) (f)[(size_t)(jgp) * np + (igp)]
#define KV(v, h, igp, jgp) (v)[((size_t)(jgp) * np + (igp)) * 2 + (h)]
#define KT(t, p, q, igp, jgp) (t)[(((size_t)(jgp) * np + (igp)) * 2 + (q)) * 2 + (p)]
#define KS2C(t, p, k, igp, jgp) (t)[(((size_t)(jgp) * np + (igp)) * 3 + (k)) * 2 + (p)]
#define KDVV(x, y) Dvv[(size_t)(y) * np + (x)]

/* K:229-269 gradient_sphere (restated only to validate the accessor macros against the pinned operator) */
void oracle_k_gradient_sphere(int np, const double *s, const double *Dvv, const double *Dinv, double rrearth,
                              double *grad) {
  double vb[2 * MAXNP * MAXNP];
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      double dsdx = 0, dsdy = 0;
      for (int kgp = 0; kgp < np; ++kgp) {
        dsdx += KDVV(jgp, kgp) * KF(s, igp, kgp); /* K:244 */
        dsdy += KDVV(jgp, kgp) * KF(s, kgp, igp); /* K:245 */
      }
      KV(vb, 0, igp, jgp) = dsdx * rrearth; /* K:247 */
      KV(vb, 1, jgp, igp) = dsdy * rrearth; /* K:248 */
    }
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(grad, 0, igp, jgp) = KT(Dinv, 0, 0, igp, jgp) * KV(vb, 0, igp, jgp) + KT(Dinv, 0, 1, igp, jgp) * KV(vb, 1, igp, jgp); /* K:260-262 */
      KV(grad, 1, igp, jgp) = KT(Dinv, 1, 0, igp, jgp) * KV(vb, 0, igp, jgp) + KT(Dinv, 1, 1, igp, jgp) * KV(vb, 1, igp, jgp); /* K:263-265 */
    }
}

/* K:271-312: grad_s += gradient_sphere(scalar) */
void oracle_gradient_sphere_update(int np, const double *s, const double *Dvv, const double *Dinv, double rrearth,
                                   double *grad) {
  double g[2 * MAXNP * MAXNP];
  oracle_k_gradient_sphere(np, s, Dvv, Dinv, rrearth, g);
  for (int q = 0; q < 2 * np * np; ++q) grad[q] += g[q]; /* K:303-308 */
}

/* K:315-358 divergence_sphere: note (dudx + dvdy) * (1/metdet * rrearth), no rmetdet array */
void oracle_k_divergence_sphere(int np, const double *v, const double *Dvv, const double *Dinv, const double *metdet,
                                double rrearth, double *div) {
  double gv[2 * MAXNP * MAXNP];
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(gv, 0, igp, jgp) = (KT(Dinv, 0, 0, igp, jgp) * KV(v, 0, igp, jgp) + KT(Dinv, 1, 0, igp, jgp) * KV(v, 1, igp, jgp)) * KF(metdet, igp, jgp); /* K:329-332 */
      KV(gv, 1, igp, jgp) = (KT(Dinv, 0, 1, igp, jgp) * KV(v, 0, igp, jgp) + KT(Dinv, 1, 1, igp, jgp) * KV(v, 1, igp, jgp)) * KF(metdet, igp, jgp); /* K:333-336 */
    }
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      double dudx = 0, dvdy = 0;
      for (int kgp = 0; kgp < np; ++kgp) {
        dudx += KDVV(jgp, kgp) * KV(gv, 0, igp, kgp); /* K:349 */
        dvdy += KDVV(igp, kgp) * KV(gv, 1, kgp, jgp); /* K:350 */
      }
      KF(div, igp, jgp) = (dudx + dvdy) * (1.0 / KF(metdet, igp, jgp) * rrearth); /* K:352-353 */
    }
}

/* K:363-403: div_v = beta * div_v + alpha * div(v) */
void oracle_divergence_sphere_update(int np, double alpha, double beta, const double *v, const double *Dvv,
                                     const double *Dinv, const double *metdet, double rrearth, double *div) {
  double d[MAXNP * MAXNP];
  oracle_k_divergence_sphere(np, v, Dvv, Dinv, metdet, rrearth, d);
  for (int q = 0; q < np * np; ++q) {
    div[q] *= beta;          /* K:398 */
    div[q] += alpha * d[q];  /* K:399 */
  }
}

/* K:452-490 vorticity_sphere_vector */
void oracle_k_vorticity_sphere_vector(int np, const double *v, const double *Dvv, const double *D, const double *metdet,
                                      double rrearth, double *vort) {
  double vc[2 * MAXNP * MAXNP];
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(vc, 0, igp, jgp) = KT(D, 0, 0, igp, jgp) * KV(v, 0, igp, jgp) + KT(D, 0, 1, igp, jgp) * KV(v, 1, igp, jgp); /* K:466-467 */
      KV(vc, 1, igp, jgp) = KT(D, 1, 0, igp, jgp) * KV(v, 0, igp, jgp) + KT(D, 1, 1, igp, jgp) * KV(v, 1, igp, jgp); /* K:468-469 */
    }
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      double dudy = 0, dvdx = 0;
      for (int kgp = 0; kgp < np; ++kgp) {
        dvdx += KDVV(jgp, kgp) * KV(vc, 1, igp, kgp); /* K:482 */
        dudy += KDVV(igp, kgp) * KV(vc, 0, kgp, jgp); /* K:483 */
      }
      KF(vort, igp, jgp) = (dvdx - dudy) * (1.0 / KF(metdet, igp, jgp) * rrearth); /* K:485-486 */
    }
}

/* K:494-534 divergence_sphere_wk */
void oracle_divergence_sphere_wk(int np, const double *v, const double *Dvv, const double *Dinv, const double *spheremp,
                                 double rrearth, double *div) {
  double gv[2 * MAXNP * MAXNP];
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(gv, 0, igp, jgp) = KT(Dinv, 0, 0, igp, jgp) * KV(v, 0, igp, jgp) + KT(Dinv, 1, 0, igp, jgp) * KV(v, 1, igp, jgp); /* K:508-509 */
      KV(gv, 1, igp, jgp) = KT(Dinv, 0, 1, igp, jgp) * KV(v, 0, igp, jgp) + KT(Dinv, 1, 1, igp, jgp) * KV(v, 1, igp, jgp); /* K:510-511 */
    }
  for (int mgp = 0; mgp < np; ++mgp)
    for (int ngp = 0; ngp < np; ++ngp) {
      double dd = 0;
      for (int jgp = 0; jgp < np; ++jgp)
        dd -= (KF(spheremp, ngp, jgp) * KV(gv, 0, ngp, jgp) * KDVV(jgp, mgp) +
               KF(spheremp, jgp, mgp) * KV(gv, 1, jgp, mgp) * KDVV(jgp, ngp)) * rrearth; /* K:525-527 */
      KF(div, ngp, mgp) = dd; /* K:529 */
    }
}

/* K:538-550 laplace_simple = divergence_sphere_wk(gradient_sphere(field)) */
void oracle_laplace_simple(int np, const double *s, const double *Dvv, const double *Dinv, const double *spheremp,
                           double rrearth, double *lap) {
  double g[2 * MAXNP * MAXNP];
  oracle_k_gradient_sphere(np, s, Dvv, Dinv, rrearth, g);            /* K:548 */
  oracle_divergence_sphere_wk(np, g, Dvv, Dinv, spheremp, rrearth, lap); /* K:549 */
}

/* K:556-596 laplace_tensor: the gradient is multiplied by tensorVisc before the weak divergence */
void oracle_laplace_tensor(int np, const double *s, const double *Dvv, const double *Dinv, const double *spheremp,
                           const double *tensorVisc, double rrearth, double *lap) {
  double g[2 * MAXNP * MAXNP], t[2 * MAXNP * MAXNP];
  oracle_k_gradient_sphere(np, s, Dvv, Dinv, rrearth, g); /* K:566 */
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(t, 0, igp, jgp) = KT(tensorVisc, 0, 0, igp, jgp) * KV(g, 0, igp, jgp) + KT(tensorVisc, 1, 0, igp, jgp) * KV(g, 1, igp, jgp); /* K:576-577 */
      KV(t, 1, igp, jgp) = KT(tensorVisc, 0, 1, igp, jgp) * KV(g, 0, igp, jgp) + KT(tensorVisc, 1, 1, igp, jgp) * KV(g, 1, igp, jgp); /* K:578-579 */
    }
  oracle_divergence_sphere_wk(np, t, Dvv, Dinv, spheremp, rrearth, lap); /* K:595 */
}

/* K:600-637 laplace_tensor_replace: "a version of laplace_tensor where input is replaced by output" — the same three
 * steps (K:609 gradient_sphere of `laplace`, K:616-619 tensorVisc times the gradient, K:636 divergence_sphere_wk back
 * into `laplace`); the whole field is read before any of it is overwritten. */
void oracle_laplace_tensor_replace(int np, const double *Dvv, const double *Dinv, const double *spheremp,
                                   const double *tensorVisc, double rrearth, double *laplace) {
  double g[2 * MAXNP * MAXNP], t[2 * MAXNP * MAXNP];
  oracle_k_gradient_sphere(np, laplace, Dvv, Dinv, rrearth, g); /* K:609 */
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(t, 0, igp, jgp) = KT(tensorVisc, 0, 0, igp, jgp) * KV(g, 0, igp, jgp) + KT(tensorVisc, 1, 0, igp, jgp) * KV(g, 1, igp, jgp); /* K:616-617 */
      KV(t, 1, igp, jgp) = KT(tensorVisc, 0, 1, igp, jgp) * KV(g, 0, igp, jgp) + KT(tensorVisc, 1, 1, igp, jgp) * KV(g, 1, igp, jgp); /* K:618-619 */
    }
  oracle_divergence_sphere_wk(np, t, Dvv, Dinv, spheremp, rrearth, laplace); /* K:636 */
}

/* K:640-690 curl_sphere_wk_testcov */
void oracle_curl_sphere_wk_testcov(int np, const double *s, const double *Dvv, const double *D, const double *mp,
                                   double rrearth, double *curls) {
  double b[2 * MAXNP * MAXNP];
  for (int q = 0; q < 2 * np * np; ++q) b[q] = 0.0; /* K:654-655 */
  for (int ngp = 0; ngp < np; ++ngp)
    for (int mgp = 0; mgp < np; ++mgp)
      for (int jgp = 0; jgp < np; ++jgp) {
        KV(b, 0, ngp, mgp) -= KF(mp, jgp, mgp) * KF(s, jgp, mgp) * KDVV(jgp, ngp); /* K:670 */
        KV(b, 1, ngp, mgp) += KF(mp, ngp, jgp) * KF(s, ngp, jgp) * KDVV(jgp, mgp); /* K:671 */
      }
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(curls, 0, igp, jgp) = (KT(D, 0, 0, igp, jgp) * KV(b, 0, igp, jgp) + KT(D, 1, 0, igp, jgp) * KV(b, 1, igp, jgp)) * rrearth; /* K:681-683 */
      KV(curls, 1, igp, jgp) = (KT(D, 0, 1, igp, jgp) * KV(b, 0, igp, jgp) + KT(D, 1, 1, igp, jgp) * KV(b, 1, igp, jgp)) * rrearth; /* K:684-686 */
    }
}

/* K:694-770 grad_sphere_wk_testcov */
void oracle_grad_sphere_wk_testcov(int np, const double *s, const double *Dvv, const double *D, const double *mp,
                                   const double *metinv, const double *metdet, double rrearth, double *grads) {
  double b[2 * MAXNP * MAXNP];
  for (int q = 0; q < 2 * np * np; ++q) b[q] = 0.0; /* K:710-711 */
  for (int ngp = 0; ngp < np; ++ngp)
    for (int mgp = 0; mgp < np; ++mgp)
      for (int jgp = 0; jgp < np; ++jgp) {
        KV(b, 0, ngp, mgp) -= (KF(mp, ngp, jgp) * KT(metinv, 0, 0, ngp, mgp) * KF(metdet, ngp, mgp) * KF(s, ngp, jgp) * KDVV(jgp, mgp) +
                               KF(mp, jgp, mgp) * KT(metinv, 0, 1, ngp, mgp) * KF(metdet, ngp, mgp) * KF(s, jgp, mgp) * KDVV(jgp, ngp)); /* K:723-735 */
        KV(b, 1, ngp, mgp) -= (KF(mp, ngp, jgp) * KT(metinv, 1, 0, ngp, mgp) * KF(metdet, ngp, mgp) * KF(s, ngp, jgp) * KDVV(jgp, mgp) +
                               KF(mp, jgp, mgp) * KT(metinv, 1, 1, ngp, mgp) * KF(metdet, ngp, mgp) * KF(s, jgp, mgp) * KDVV(jgp, ngp)); /* K:738-750 */
      }
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp) {
      KV(grads, 0, igp, jgp) = (KT(D, 0, 0, igp, jgp) * KV(b, 0, igp, jgp) + KT(D, 1, 0, igp, jgp) * KV(b, 1, igp, jgp)) * rrearth; /* K:761-763 */
      KV(grads, 1, igp, jgp) = (KT(D, 0, 1, igp, jgp) * KV(b, 0, igp, jgp) + KT(D, 1, 1, igp, jgp) * KV(b, 1, igp, jgp)) * rrearth; /* K:764-766 */
    }
}

/* K:849-915 vlaplace_sphere_wk_cartesian_reduced (UNDAMPRRCART defined at K:891: the rigid-rotation term is
 * kept); undamp_rr == 0 gives K:777-844 vlaplace_sphere_wk_cartesian (no such term). */
void oracle_vlaplace_sphere_wk_cartesian(int np, const double *v, const double *Dvv, const double *Dinv,
                                         const double *spheremp, const double *tensorVisc, const double *vec_sph2cart,
                                         double rrearth, int undamp_rr, double *lap) {
  double comp[3][MAXNP * MAXNP], l[3][MAXNP * MAXNP];
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp)
      for (int k = 0; k < 3; ++k)
        KF(comp[k], igp, jgp) = KS2C(vec_sph2cart, 0, k, igp, jgp) * KV(v, 0, igp, jgp) + KS2C(vec_sph2cart, 1, k, igp, jgp) * KV(v, 1, igp, jgp); /* K:869-875 */
  for (int k = 0; k < 3; ++k) oracle_laplace_tensor(np, comp[k], Dvv, Dinv, spheremp, tensorVisc, rrearth, l[k]); /* K:880-882 */
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp)
      for (int h = 0; h < 2; ++h) {
        double r = KS2C(vec_sph2cart, h, 0, igp, jgp) * KF(l[0], igp, jgp) + KS2C(vec_sph2cart, h, 1, igp, jgp) * KF(l[1], igp, jgp) +
                   KS2C(vec_sph2cart, h, 2, igp, jgp) * KF(l[2], igp, jgp); /* K:893-895 */
        if (undamp_rr) r = r + 2.0 * KF(spheremp, igp, jgp) * KV(v, h, igp, jgp) * rrearth * rrearth; /* K:896-897 */
        KV(lap, h, igp, jgp) = r;
      }
}

/* K:938-993 vlaplace_sphere_wk_contra */
void oracle_vlaplace_sphere_wk_contra(int np, const double *v, const double *Dvv, const double *D, const double *Dinv,
                                      const double *mp, const double *spheremp, const double *metinv, const double *metdet,
                                      double nu_ratio, double rrearth, double *lap) {
  double div[MAXNP * MAXNP], vort[MAXNP * MAXNP], gradcov[2 * MAXNP * MAXNP], curlcov[2 * MAXNP * MAXNP];
  oracle_k_divergence_sphere(np, v, Dvv, Dinv, metdet, rrearth, div);      /* K:959 */
  oracle_k_vorticity_sphere_vector(np, v, Dvv, D, metdet, rrearth, vort);  /* K:960 */
  for (int q = 0; q < np * np; ++q) div[q] *= nu_ratio;                    /* K:967 */
  oracle_grad_sphere_wk_testcov(np, div, Dvv, D, mp, metinv, metdet, rrearth, gradcov); /* K:973 */
  oracle_curl_sphere_wk_testcov(np, vort, Dvv, D, mp, rrearth, curlcov);                 /* K:974 */
  for (int igp = 0; igp < np; ++igp)
    for (int jgp = 0; jgp < np; ++jgp)
      for (int h = 0; h < 2; ++h) {
        double r = 2.0 * KF(spheremp, igp, jgp) * KV(v, h, igp, jgp) * rrearth * rrearth; /* K:982-986 */
        r += KV(gradcov, h, igp, jgp) - KV(curlcov, h, igp, jgp);                         /* K:988-989 */
        KV(lap, h, igp, jgp) = r;
      }
}

/* EulerStepFunctor.hpp:32-68 (E:), one element: for every tracer and level  v_buf = vstar * qdp (E:59-60),
 * q_buf = qdp (E:61), divergence_sphere_update(-dt, 1.0, ..., v_buf, q_buf) (E:65-66).  The reference cannot compile
 * the functor (8 arguments for the 9 parameters of K:363-370): PARITY UNPINNED, this is what it states.
 * vstar [nlev][np][np][2]; qdp [qsize_d][2][nlev][np][np] (the element's slice of state_Qdp); qtens [qsize][nlev][np][np]. */
void oracle_euler_step(int np, int nlev, int qsize, int qn0, double dt, const double *vstar, const double *qdp,
                       const double *Dvv, const double *Dinv, const double *metdet, double rrearth, double *qtens) {
  const int pp = np * np;
  double v_buf[2 * MAXNP * MAXNP];
  for (int iq = 0; iq < qsize; ++iq)
    for (int ilev = 0; ilev < nlev; ++ilev) {
      const double *q = qdp + ((size_t)(iq * 2 + qn0) * nlev + ilev) * pp;
      const double *vs = vstar + (size_t)ilev * pp * 2;
      double *q_buf = qtens + ((size_t)iq * nlev + ilev) * pp;
      for (int idx = 0; idx < pp; ++idx) {
        v_buf[2 * idx + 0] = vs[2 * idx + 0] * q[idx]; /* E:59 */
        v_buf[2 * idx + 1] = vs[2 * idx + 1] * q[idx]; /* E:60 */
        q_buf[idx] = q[idx];                           /* E:61 */
      }
      oracle_divergence_sphere_update(np, -dt, 1.0, v_buf, Dvv, Dinv, metdet, rrearth, q_buf); /* E:65-66 */
    }
}
