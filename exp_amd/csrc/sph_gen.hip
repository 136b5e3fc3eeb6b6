// Spherical force method at ANY harmonic order: run-time loops over (l, m) where sph_kernels.h unrolls them at
// compile time for lmax <= SPH_MAX_L.  The reference takes whatever `Lmax` the YAML gives it (src/Sphere.cc:28-96,
// src/SphericalBasis.cc:96-121); a drop-in must not refuse lmax = 16 because its fast kernels stop at 12.  These
// kernels are the plain formulation -- one particle per lane, every moment an atomic, every table row a gather -- and
// run at a fraction of the unrolled kernels' speed; they are what exp_amd_sph_create selects above SPH_MAX_L (and, for
// tests, at any order with EXP_AMD_SPH_GENERIC=1: same moments, same projected table, same results to rounding).
//   k_sph_acc_gen   : W[cell][row][2] += -4 pi m P0 x_k Yh_row   (determine_coefficients_thread, src/SphericalBasis.cc:
//                     429-599; a slot range, or the LIST mode of the level-change differencing)
//   k_sph_upd_gen   : the per-particle differencing / sparse accumulation (k_sph_mstep_update; multistep_update,
//                     src/SphericalBasis.cc:1156-1228)
//   k_sph_force_gen : the general evaluation (sph_force_chunk<LMAX, 0> + sph_field; determine_acceleration_and_
//                     potential_thread, src/SphericalBasis.cc:1476-1660), far extrapolation inline
// The rescaled Legendre recurrence (sph_kernels.h: lc_s, lc_a, lc_c, lc_E) takes its constants from SphDev::gen_ac /
// gen_e, filled by the host from the same constexpr functions the unrolled kernels fold into literals.
#include "sph_kernels.h"

// ---- accumulation over a slot range or a list of movers -------------------------------------------------------------
template <bool LIST>
__global__ void __launch_bounds__(256)
k_sph_acc_gen(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
              const double *__restrict__ M, const uint32_t *__restrict__ lev_off, int lo, int hi,
              double *__restrict__ W, unsigned long long *__restrict__ used_out, int multilevel, AccList al)
{
  const int L = S.lmax;
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  const size_t i = beg + (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool have = i < end;
  double x = 0, y = 0, z = 0, m = 0;
  int ca = 0;
  if (have) {
    if constexpr (LIST) acc_list_fetch(al, X, Y, Z, M, S.umass, i, x, y, z, m, ca);
    else {
      x = X[i]; y = Y[i]; z = Z[i]; m = S.umass != 0.0 ? S.umass : M[i];
      if (multilevel) {
        int lv = lo;
        while (lv < hi && i >= lev_off[lv + 1]) lv++;
        ca = lv * (S.numr - 1);
      }
    }
  }
  const AccIn in = sph_acc_input<LIST>(S, (ldp) nullptr, x, y, z, m, have && ca >= 0, ca < 0 ? 0 : ca);
  const bool on = in.idx >= 0;
  if (!LIST) {
    const unsigned long long b = __ballot(on);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(used_out, (unsigned long long)__popcll(b));
  }
  if (!__any(on)) return;
  double *w = W + (size_t)(on ? in.idx : 0) * S.nrows * 2;
  double pmm = S.gen_e[0];
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  for (int mm = 0; mm <= L; mm++) {
    if (mm == 1) { pmm *= S.gen_e[1] * in.sinth; cm = in.cphi; sm = in.sphi; }
    else if (mm > 1) {
      pmm *= S.gen_e[mm] * in.sinth;
      const double cn = 2.0 * in.cphi * cm - cm1, sn = 2.0 * in.cphi * sm - sm1;      // src/Basis.cc:107-110
      cm1 = cm; sm1 = sm; cm = cn; sm = sn;
    }
    if (mm > 0 && S.M0_acc) break;
    double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
    for (int l = mm; l <= L; l++) {
      double plm;
      if (l == mm) plm = pmm;
      else if (l == mm + 1) plm = S.gen_ac[((size_t)l * (L + 1) + mm) * 2] * tprev;
      else plm = fma(S.gen_ac[((size_t)l * (L + 1) + mm) * 2], tprev, -pl2);
      tprev = in.costh * plm;
      pl2 = pl1;
      pl1 = plm;
      if (on) {
        const int row = l * l + (mm ? 2 * mm - 1 : 0);
        const double yc = mm == 0 ? plm : plm * cm;
        unsafeAtomicAdd(w + 2 * row, det_round(in.a1 * yc, S.detC));
        unsafeAtomicAdd(w + 2 * row + 1, det_round(in.a2 * yc, S.detC));
        if (mm > 0) {
          const double ys = plm * sm;
          unsafeAtomicAdd(w + 2 * row + 2, det_round(in.a1 * ys, S.detC));
          unsafeAtomicAdd(w + 2 * row + 3, det_round(in.a2 * ys, S.detC));
        }
      }
    }
  }
}

// ---- per-particle differencing / sparse accumulation (k_sph_mstep_update, statement for statement) ------------------
__global__ void __launch_bounds__(256)
k_sph_upd_gen(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
              const double *__restrict__ M, const uint8_t *__restrict__ lev, const uint8_t *__restrict__ newlev,
              const uint32_t *__restrict__ lev_off, int first, int last, int mfirst, double *__restrict__ Wd, int plain,
              unsigned long long *__restrict__ used_out, const uint32_t *__restrict__ list)
{
  const int L = S.lmax;
  size_t i = 0;
  bool have = false;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (list) {
    if (g < lev_off[1]) { i = list[g]; have = true; }
  } else {
    i = lev_off[first] + g;
    have = i < lev_off[last + 1];
  }
  bool mover = false;
  int from = 0, to = 0;
  if (have) {
    from = lev[i];
    to = plain ? from : newlev[i];
    mover = plain || from != to;
  }
  if (!__any(mover)) return;
  double xx = 0, yy = 0, zz = 1, mass = 0;
  if (mover) {
    xx = X[i] - S.cx; yy = Y[i] - S.cy; zz = Z[i] - S.cz; mass = M[i];
    if (SPH_FRZ_ON(S) && sph_frozen(S, X[i], Y[i], Z[i])) mover = false;
  }
  const double r = sqrt(sq_add_lit(sq_sum2_lit(xx, yy), zz)) + DSMALL;       // (every product rounded on its own: sq_sum2_lit)
  if (plain) {
    if (!(r >= S.rmin && r <= S.rmax)) mover = false;
    const unsigned long long in = __ballot(mover);
    if ((threadIdx.x & 63) == 0 && in) atomicAdd(used_out, (unsigned long long)__popcll(in));
  } else if (!(r < S.rmax)) mover = false;
  if (!__any(mover)) return;
  const double costh = zz / r;
  double cphi, sphi;
  phi_trig(xx, yy, cphi, sphi);
  const double xi = sph_r_to_xi(S, r / S.scale);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
  const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const double t0 = mass * S.fac0 * P0;
  const double a1 = t0 * x1, a2 = t0 * x2;
  const size_t wl = (size_t)(S.numr - 1) * S.nrows * 2;
  double *wto = Wd + (size_t)to * wl + (size_t)idx * S.nrows * 2;
  double *wfr = Wd + (size_t)from * wl + (size_t)idx * S.nrows * 2;
  const bool sub = !plain && mover && from >= mfirst;
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  double pmm = S.gen_e[0];
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  for (int m = 0; m <= L; m++) {
    if (m == 1) { pmm *= S.gen_e[1] * somx2; cm = cphi; sm = sphi; }
    else if (m > 1) {
      pmm *= S.gen_e[m] * somx2;
      const double cn = 2.0 * cphi * cm - cm1, sn = 2.0 * cphi * sm - sm1;
      cm1 = cm; sm1 = sm; cm = cn; sm = sn;
    }
    double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
    for (int l = m; l <= L; l++) {
      double plm;
      if (l == m) plm = pmm;
      else if (l == m + 1) plm = S.gen_ac[((size_t)l * (L + 1) + m) * 2] * tprev;
      else plm = fma(S.gen_ac[((size_t)l * (L + 1) + m) * 2], tprev, -pl2);
      tprev = costh * plm;
      pl2 = pl1;
      pl1 = plm;
      if (mover) {
        const int row = l * l + (m ? 2 * m - 1 : 0);
        const double yc = (m == 0) ? plm : plm * cm;
        const double v1 = det_round(a1 * yc, S.detC), v2 = det_round(a2 * yc, S.detC);
        unsafeAtomicAdd(wto + 2 * row, v1);
        unsafeAtomicAdd(wto + 2 * row + 1, v2);
        if (sub) { unsafeAtomicAdd(wfr + 2 * row, -v1); unsafeAtomicAdd(wfr + 2 * row + 1, -v2); }
        if (m > 0) {
          const double ys = plm * sm;
          const double u1 = det_round(a1 * ys, S.detC), u2 = det_round(a2 * ys, S.detC);
          unsafeAtomicAdd(wto + 2 * row + 2, u1);
          unsafeAtomicAdd(wto + 2 * row + 3, u2);
          if (sub) { unsafeAtomicAdd(wfr + 2 * row + 2, -u1); unsafeAtomicAdd(wfr + 2 * row + 3, -u2); }
        }
      }
    }
  }
}

// ---- the general evaluation (sph_field with run-time loops; LIT lanes inline) ---------------------------------------
__device__ __forceinline__ ForceOut
sph_field_gen(const SphDev &S, double costh, double xc, double cphi, double sphi, const double *__restrict__ t4,
              double x2, double pf, bool ioff, double rr, double kappa0, double pf_lit)
{
  const int L = S.lmax;
  ForceOut o{0.0, 0.0, 0.0, 0.0};
  const bool lit = !ioff && (pf_lit < S.lit_lo || pf_lit > S.lit_hi);
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  double pmm = S.gen_e[0];
  double cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
  const bool clamped = 1.0 - fabs(costh) < SPH_POLAR_FAC;      // (see leg0_lit_step, sph_kernels.h)
  double lp1 = 0.0, lp2 = 0.0;
  int qb = 0;                                          // first slot of this m (gen_t4_base)
  for (int m = 0; m <= L; m++) {
    if (m == 1) { pmm *= S.gen_e[1] * somx2; cm = cphi; sm = sphi; }
    else if (m > 1) {
      pmm *= S.gen_e[m] * somx2;
      const double cn = 2.0 * cphi * cm - cm1, sn = 2.0 * cphi * sm - sm1;
      cm1 = cm; sm1 = sm; cm = cn; sm = sn;
    }
    if (m == 1) qb += (L + 1) + ((L + 1) & 1);
    else if (m > 1) qb += 2 * (L - (m - 1) + 1);
    bool m_on = true;
    if (S.EVEN_M && (m & 1)) m_on = false;
    if (S.M0_only && m != 0) m_on = false;
    double rl = 1.0;                                   // (rmax/r0)^(l+1), from l = m  (src/SphericalBasis.cc:1605-1628)
    if (ioff) { rl = rr; for (int k = 0; k < m; k++) rl *= rr; }
    double Al = 0.0, Bl = 0.0, Ar = 0.0, Br = 0.0, At = 0.0, Bt = 0.0;
    double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
    for (int l = m; l <= L; l++) {
      const double *ac = S.gen_ac + ((size_t)l * (L + 1) + m) * 2;
      double plm, qlm;
      if (l == m) plm = pmm;
      else if (l == m + 1) plm = ac[0] * tprev;
      else plm = fma(ac[0], tprev, -pl2);
      tprev = costh * plm;
      if (l == m) qlm = (xc * plm) * l;
      else qlm = fma((double)l, xc * plm, -(ac[1] * pl1));
      if (m == 0 && clamped) {          // (leg0_lit_step, sph_kernels.h: the reference's own m = 0 recurrence near the poles)
        double ql;
        leg0_lit_step(l, costh, xc, lp1, lp2, ql);
        qlm = ql * (plm / lp1);
      }
      pl2 = pl1;
      pl1 = plm;
      const int slot = qb + (m == 0 ? (l - m) : 2 * (l - m));
      const int q = 4 * slot;
      bool on = m_on;
      if (l == 0 && S.NO_L0) on = false;
      if (l == 1 && S.NO_L1) on = false;
      if (l > 0 && S.EVEN_L && (l & 1)) on = false;
      if (on) {
        double pc = fma(x2, t4[q + 1], t4[q + 0]);
        double dpc = fma(pf, t4[q + 3], t4[q + 2]);
        if (lit) dpc = sph_dp_lit(S, slot, l, pf_lit);
        pc *= rl;
        dpc = ioff ? (kappa0 * (l + 1)) * pc : dpc;
        Al = fma(plm, pc, Al);
        Ar = fma(plm, dpc, Ar);
        At = fma(qlm, pc, At);
        if (m > 0) {
          double ps = fma(x2, t4[q + 5], t4[q + 4]);
          double dps = fma(pf, t4[q + 7], t4[q + 6]);
          if (lit) dps = sph_dp_lit(S, slot + 1, l, pf_lit);
          ps *= rl;
          dps = ioff ? (kappa0 * (l + 1)) * ps : dps;
          Bl = fma(plm, ps, Bl);
          Br = fma(plm, dps, Br);
          Bt = fma(qlm, ps, Bt);
        }
      }
      rl *= ioff ? rr : 1.0;
    }
    if (m == 0) { o.potl += Al; o.potr += Ar; o.pott += At; }
    else {
      o.potl += Al * cm + Bl * sm;
      o.potr += Ar * cm + Br * sm;
      o.pott += At * cm + Bt * sm;
      o.potp += (Bl * cm - Al * sm) * m;
    }
  }
  return o;
}

__global__ void __launch_bounds__(256)
k_sph_force_gen(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi, const double *__restrict__ T4,
                double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT,
                double *__restrict__ VX, double *__restrict__ VY, double *__restrict__ VZ, double dt_kick, int assign,
                uint32_t *__restrict__ key_out, double nk_dtk, double nk_dtd, int store_v,
                uint32_t *__restrict__ nwork_clear)
{
  if (nwork_clear && blockIdx.x == 0 && threadIdx.x == 0) *nwork_clear = 0u;
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  const size_t i = beg + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= end) return;
  // sph_force_chunk<LMAX, 0>, statement for statement (src/SphericalBasis.cc:1545-1560)
  const double px = X[i], py = Y[i], pz = Z[i];
  const double xx = px - S.cx, yy = py - S.cy, zz = pz - S.cz;
  const double fac = sq_sum2_lit(xx, yy);           // (no fused multiply-add: see sq_sum2_lit)
  double r = sqrt(sq_add_lit(fac, zz)) + S.dsmall;
  const double costh = zz / r;
  double cphi, sphi;
  phi_trig(xx, yy, cphi, sphi);
  bool ioff = false;
  const double r0 = r;
  if (r > S.rmax && !S.no_exterior) { ioff = true; r = S.rmax; }
  const double rs = r / S.scale;
  const double xi = sph_r_to_xi(S, rs);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
  const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const int jdx = idx < 1 ? 1 : idx;
  const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
  const double ffac = sph_d_xi_to_r(S, xi) * S.inv_dxi;
  double xc = costh;
  if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
  const double dfac = 1.0 / sq_add_lit(-1.0, xc);
  const double rr = S.rmax / r0;
  const double kappa0 = -P0 / (r0 * ffac);
  const double *t4 = T4 + (size_t)idx * 4 * S.trows;
  const ForceOut o = sph_field_gen(S, costh, xc, cphi, sphi, t4, x2, pf, ioff, rr, kappa0, (xi - S.xi[jdx]) / S.dxi);
  sph_force_finish<false>(S, o, i, xx, yy, zz, px, py, pz, fac, 1.0 / r, 1.0 / fac, P0, ffac, dfac, AX, AY, AZ, POT, VX, VY,
                          VZ, dt_kick, assign, key_out, nk_dtk, nk_dtd, store_v);
}

// ---- launchers with the signatures of the per-LMAX ones (sph_inst.hip) ----------------------------------------------
void expamd_sph_acc_gen(const SphAccArgs &a)
{
  if (a.list) {
    if (a.n == 0) return;
    const AccList al{a.list, a.lev, a.newlev, a.mfirst, a.S.numr - 1, a.nslices > 2 ? 1 : 0};
    k_sph_acc_gen<true><<<dim3(cdiv(a.n, 256), 1, a.nslices), 256, 0, a.stream>>>(
        a.S, a.X, a.Y, a.Z, a.M, a.lev_off, 0, 0, a.W, a.used, 1, al);
    return;
  }
  size_t n = 0;
  if (a.counts) for (int j = 0; j <= a.hi - a.lo; j++) n += a.counts[j]; else n = a.n;
  if (n == 0) return;
  k_sph_acc_gen<false><<<cdiv(n, 256), 256, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off, a.lo, a.hi, a.W, a.used,
                                                            a.wlevels ? 1 : 0, AccList{});
}

void expamd_sph_upd_gen(const SphUpdArgs &a)
{
  if (a.n == 0) return;
  k_sph_upd_gen<<<cdiv(a.n, 256), 256, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev, a.newlev, a.lev_off, a.first, a.last,
                                                      a.mfirst, a.Wd, a.plain, a.used, a.list);
}

void expamd_sph_force_gen(const SphForceArgs &a)
{
  if (a.n == 0) return;
  ProfScope ps(a.ctx, "k_sph_force_general");
  k_sph_force_gen<<<cdiv(a.n, 256), 256, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.T4, a.AX, a.AY, a.AZ,
                                                        a.POT, a.VX, a.VY, a.VZ, a.dt_kick, a.assign, a.key_out, a.nk_dtk,
                                                        a.nk_dtd, a.store_v, a.nwork_next);
}

// ---- thin active sets, second formulation (round 4): one WAVE per particle for the forces, 64-particle tiles with level
// runs for the accumulation -- any order, no per-LMAX instantiation ------------------------------------------------------
// k_sph_force_wave: the lanes of a wave own the slots of the projected table (lane q, q + 64, ...): each forms ITS row of
// T4 for the particle's cell straight from E and the coefficient set (three 'G' sums over n: k_sph_project + k_sph_project4
// through sph_t4_entry), runs the rescaled recurrence up to its own (l, m) (at most lmax steps), multiplies, and the four
// field sums are reduced over the wave.  The scalar part (radius, cell, weights, the exterior continuation and the pole
// clamp of the reference, src/SphericalBasis.cc:1545-1660) is the general evaluation's, computed redundantly by every lane:
// no LDS, no barrier, no staging buffer, and a few hundred particles already fill the GPU with waves.
__global__ void __launch_bounds__(256)
k_sph_force_wave(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
                 const uint32_t *__restrict__ lev_off, int lev_lo, int lev_hi, const double *__restrict__ coef,
                 double *__restrict__ AX, double *__restrict__ AY, double *__restrict__ AZ, double *__restrict__ POT,
                 double *__restrict__ VX, double *__restrict__ VY, double *__restrict__ VZ, int assign)
{
  const int L = S.lmax, lane = threadIdx.x & 63;
  const int lsn = (L + 1) * S.nmax;
  const size_t beg = lev_off[lev_lo], end = lev_off[lev_hi + 1];
  for (size_t i = beg + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < end; i += (size_t)gridDim.x * 4) {
    // ---- sph_force_chunk<LMAX, 0>'s prologue, the same for every lane
    const double px = X[i], py = Y[i], pz = Z[i];
    const double xx = px - S.cx, yy = py - S.cy, zz = pz - S.cz;
    const double fac = sq_sum2_lit(xx, yy);
    double r = sqrt(sq_add_lit(fac, zz)) + S.dsmall;
    const double costh = zz / r;
    double cphi, sphi;
    phi_trig(xx, yy, cphi, sphi);
    bool ioff = false;
    const double r0 = r;
    if (r > S.rmax && !S.no_exterior) { ioff = true; r = S.rmax; }
    const double xi = sph_r_to_xi(S, r / S.scale);
    const int idx = sph_cell(S, xi);
    const double x1 = (S.xi[idx + 1] - xi) * S.inv_dxi;
    const double x2 = (xi - S.xi[idx]) * S.inv_dxi;
    const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
    const int jdx = idx < 1 ? 1 : idx;
    const double pf = (xi - S.xi[jdx]) * S.inv_dxi;
    const double ffac = sph_d_xi_to_r(S, xi) * S.inv_dxi;
    double xc = costh;
    if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
    const double dfac = 1.0 / sq_add_lit(-1.0, xc);
    const double rr = S.rmax / r0;
    const double kappa0 = -P0 / (r0 * ffac);
    const double pf_lit = (xi - S.xi[jdx]) / S.dxi;
    const bool lit = !ioff && (pf_lit < S.lit_lo || pf_lit > S.lit_hi);
    const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
    const double pa = S.p0[jdx - 1], pb = S.p0[jdx], pc_ = S.p0[jdx + 1];
    double potl = 0.0, potr = 0.0, pott = 0.0, potp = 0.0;
    for (int q = lane; q < S.trows; q += 64) {
      const int row = S.lit_rowmap[q];              // coefficient row of this slot; < 0: switched off, or the pad row
      if (row < 0) continue;
      const int l = S.gen_slot[2 * q], mc = S.gen_slot[2 * q + 1], m = mc & 0x7f, cs = mc >> 7;
      // ---- this slot of T4 for the particle's cell: G at the three nodes of the force stencil
      const double *e = S.E + (size_t)(jdx - 1) * lsn + l * S.nmax;
      const double *c = coef + (size_t)row * S.nmax;
      double ga = 0.0, gb = 0.0, gc = 0.0;
      for (int n0 = 0; n0 < S.nmax; n0 += 6) {
        double cv[6], ea[6], eb[6], ec[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const bool in = n0 + u < S.nmax;
          cv[u] = in ? c[n0 + u] : 0.0;
          ea[u] = in ? e[n0 + u] : 0.0;
          eb[u] = in ? e[lsn + n0 + u] : 0.0;
          ec[u] = in ? e[2 * lsn + n0 + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
          if (n0 + u < S.nmax) { ga = fma(ea[u], cv[u], ga); gb = fma(eb[u], cv[u], gb); gc = fma(ec[u], cv[u], gc); }
      }
      const double g0 = idx < 1 ? ga : gb, g1 = idx < 1 ? gb : gc;
      double t0, t1, t2, t3;
      sph_t4_entry(S.lit_tscale[q], g0, g1, pa * ga, pb * gb, pc_ * gc, t0, t1, t2, t3);
      // ---- Ph(l, m), (x^2 - 1) dPh(l, m) and the trig factor of this slot
      double pmm = S.gen_e[0], cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
      for (int k = 1; k <= m; k++) {
        pmm *= S.gen_e[k] * somx2;
        if (k == 1) { cm = cphi; sm = sphi; }
        else {
          const double cn = 2.0 * cphi * cm - cm1, sn = 2.0 * cphi * sm - sm1;
          cm1 = cm; sm1 = sm; cm = cn; sm = sn;
        }
      }
      double plm = pmm, pl1 = 0.0, pl2 = 0.0, tprev = costh * pmm, qlm = (xc * pmm) * m;
      for (int k = m + 1; k <= l; k++) {
        const double *ac = S.gen_ac + ((size_t)k * (L + 1) + m) * 2;
        pl2 = pl1; pl1 = plm;
        plm = (k == m + 1) ? ac[0] * tprev : fma(ac[0], tprev, -pl2);
        tprev = costh * plm;
        qlm = fma((double)k, xc * plm, -(ac[1] * pl1));
      }
      if (m == 0 && 1.0 - fabs(costh) < SPH_POLAR_FAC) {      // (leg0_lit_step: the reference's own m = 0 recurrence near the poles)
        double lp1 = 0.0, lp2 = 0.0, ql = 0.0;
        for (int k = 0; k <= l; k++) leg0_lit_step(k, costh, xc, lp1, lp2, ql);
        qlm = ql * (plm / lp1);
      }
      double rl = 1.0;
      if (ioff) { rl = rr; for (int k = 0; k < l; k++) rl *= rr; }      // (rmax / r0)^(l + 1)
      double pcv = fma(x2, t1, t0) * rl;
      double dpc = fma(pf, t3, t2);
      if (lit) dpc = sph_dp_lit(S, q, l, pf_lit);
      dpc = ioff ? (kappa0 * (l + 1)) * pcv : dpc;
      const double trig = cs ? sm : cm;
      potl = fma(plm * pcv, trig, potl);
      potr = fma(plm * dpc, trig, potr);
      pott = fma(qlm * pcv, trig, pott);
      if (m) potp += (cs ? plm * pcv * cm : -(plm * pcv * sm)) * m;
    }
    for (int off = 32; off > 0; off >>= 1) {
      potl += __shfl_xor(potl, off); potr += __shfl_xor(potr, off);
      pott += __shfl_xor(pott, off); potp += __shfl_xor(potp, off);
    }
    if (lane == 0) {
      const ForceOut o{potl, potr, pott, potp};
      sph_force_finish<false>(S, o, i, xx, yy, zz, px, py, pz, fac, 1.0 / r, 1.0 / fac, P0, ffac, dfac, AX, AY, AZ, POT, VX, VY,
                              VZ, 0.0, assign, nullptr, 0.0, 0.0, 1);
    }
  }
}

// k_sph_acc_tile: a tile of `tile` particles per block pass.  Lane t of the first wave prepares particle t (window, cell,
// weights: sph_acc_input; its (lmax+1)^2 rescaled harmonics into LDS); the block forms pe[p][l][n] = a1 E[i][l][n] +
// a2 E[i+1][l][n] with coalesced reads; each thread then owns a few coefficients (row, n) and sums yv[p][row] pe[p][l][n]
// over the tile's RUNS of equal level (the range is level-contiguous: a handful of runs), one atomic per run into
// part[level - lo][seg][row][n] (k_sph_contract's layout: the summing kernels finish the job and leave it zero).
#define SPH_TILE_MAX 64
#define SPH_TILE_KMAX 16          // coefficients per thread at most: ncoef <= 4096 (SphForce::thin_ok)
__global__ void __launch_bounds__(256)
k_sph_acc_tile(SphDev S, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z,
               const double *__restrict__ M, const uint32_t *__restrict__ lev_off, int lo, int hi,
               const double *__restrict__ wscale, double *__restrict__ part, unsigned long long *__restrict__ used_out,
               int tile)
{
  extern __shared__ __attribute__((aligned(16))) double tile_lds[];
  __shared__ int s_idx[SPH_TILE_MAX], s_run_beg[20], s_run_lev[20], s_nrun;
  __shared__ double s_a1[SPH_TILE_MAX], s_a2[SPH_TILE_MAX];
  const int L = S.lmax, nrows = S.nrows, lsn = (L + 1) * S.nmax, ncoef = nrows * S.nmax;
  const int yst = nrows | 1;                                  // odd row stride: lanes of phase A write distinct banks
  double *yv = tile_lds;                                      // [tile][yst]
  double *pe = tile_lds + (((size_t)tile * yst + 1) & ~(size_t)1);     // [tile][lsn]
  const size_t beg = lev_off[lo], end = lev_off[hi + 1];
  const int seg = blockIdx.x % CSEG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // the coefficients this thread owns (fixed over the tiles): k = t, t + 256, ...
  int krow[SPH_TILE_KMAX], kln[SPH_TILE_KMAX], nk = 0;
  double kws[SPH_TILE_KMAX];
  for (int k = t; k < ncoef && nk < SPH_TILE_KMAX; k += 256, nk++) {
    const int row = k / S.nmax;
    krow[nk] = row;
    kln[nk] = sph_l_of_row(row) * S.nmax + (k - row * S.nmax);
    kws[nk] = wscale[row];
  }
  const bool um = S.umass != 0.0;
  for (size_t base = beg + (size_t)blockIdx.x * tile; base < end; base += (size_t)gridDim.x * tile) {
    const int np = (int)((end - base) < (size_t)tile ? (end - base) : (size_t)tile);
    if (wave == 0) {
      const size_t i = base + lane;
      const bool valid = lane < np;
      double x = 0, y = 0, z = 0, m = 0;
      if (valid) { x = X[i]; y = Y[i]; z = Z[i]; m = um ? S.umass : M[i]; }
      const AccIn in = sph_acc_input<false>(S, (ldp) nullptr, x, y, z, m, valid);
      int lv = lo;
      while (lv < hi && i >= lev_off[lv + 1]) lv++;
      if (!valid) lv = -1;
      // runs of equal level among the tile's particles (lane order = slot order)
      const int prev = __shfl_up(lv, 1);
      const unsigned long long starts = __ballot(valid && (lane == 0 || lv != prev));
      if (valid && (lane == 0 || lv != prev)) {
        const int r = __popcll(starts & ((1ull << lane) - 1ull));
        if (r < 20) { s_run_beg[r] = lane; s_run_lev[r] = lv; }
      }
      const unsigned long long inwin = __ballot(in.idx >= 0);
      if (lane == 0) {
        s_nrun = min(20, (int)__popcll(starts));
        if (inwin) atomicAdd(used_out, (unsigned long long)__popcll(inwin));
      }
      if (lane < tile) {
        s_idx[lane] = in.idx; s_a1[lane] = in.a1; s_a2[lane] = in.a2;
        double *yr = yv + (size_t)lane * yst;
        const bool on = in.idx >= 0;
        double pmm = S.gen_e[0], cm = 1.0, sm = 0.0, cm1 = 1.0, sm1 = 0.0;
        for (int mm = 0; mm <= L; mm++) {
          if (mm == 1) { pmm *= S.gen_e[1] * in.sinth; cm = in.cphi; sm = in.sphi; }
          else if (mm > 1) {
            pmm *= S.gen_e[mm] * in.sinth;
            const double cn = 2.0 * in.cphi * cm - cm1, sn = 2.0 * in.cphi * sm - sm1;
            cm1 = cm; sm1 = sm; cm = cn; sm = sn;
          }
          const bool m_on = on && (mm == 0 || !S.M0_acc);
          double pl2 = 0.0, pl1 = 0.0, tprev = 0.0;
          for (int l = mm; l <= L; l++) {
            double plm;
            if (l == mm) plm = pmm;
            else if (l == mm + 1) plm = S.gen_ac[((size_t)l * (L + 1) + mm) * 2] * tprev;
            else plm = fma(S.gen_ac[((size_t)l * (L + 1) + mm) * 2], tprev, -pl2);
            tprev = in.costh * plm;
            pl2 = pl1;
            pl1 = plm;
            const int row = l * l + (mm ? 2 * mm - 1 : 0);
            if (mm == 0) yr[row] = m_on ? plm : 0.0;
            else { yr[row] = m_on ? plm * cm : 0.0; yr[row + 1] = m_on ? plm * sm : 0.0; }
          }
        }
      }
    }
    __syncthreads();
    // pe: wave w takes particles w, w + 4, ...; four at a time so that eight loads per lane are in flight
    for (int p0 = wave * 4; p0 < np; p0 += 16) {
      for (int k0 = lane; k0 < lsn; k0 += 64) {
        double ea[4], eb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int p = p0 + u;
          ea[u] = eb[u] = 0.0;
          if (p < np) {
            const int idx = s_idx[p];
            if (idx >= 0) { ea[u] = S.E[(size_t)idx * lsn + k0]; eb[u] = S.E[(size_t)(idx + 1) * lsn + k0]; }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int p = p0 + u;
          if (p < np) pe[(size_t)p * lsn + k0] = fma(s_a2[p], eb[u], s_a1[p] * ea[u]);
        }
      }
    }
    __syncthreads();
    const int nrun = s_nrun;
    for (int j = 0; j < nk; j++) {
      const int row = krow[j], ln = kln[j];
      const int k = t + 256 * j;
      for (int r = 0; r < nrun; r++) {
        const int pb = s_run_beg[r], pe_ = r + 1 < nrun ? s_run_beg[r + 1] : np;
        double acc = 0.0;
        for (int p0 = pb; p0 < pe_; p0 += 8) {
          double yy_[8], pp_[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const bool in = p0 + u < pe_;
            yy_[u] = in ? yv[(size_t)(p0 + u) * yst + row] : 0.0;
            pp_[u] = in ? pe[(size_t)(p0 + u) * lsn + ln] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) acc = fma(yy_[u], pp_[u], acc);
        }
        if (acc != 0.0) unsafeAtomicAdd(part + ((size_t)(s_run_lev[r] - lo) * CSEG + seg) * ncoef + k, acc * kws[j]);
      }
    }
    __syncthreads();
  }
}

void expamd_sph_thin_force_gen(const SphThinForceArgs &a)
{
  if (a.n == 0) return;
  size_t grid = cdiv(a.n, 4);
  if (grid > 16384) grid = 16384;
  k_sph_force_wave<<<(unsigned)grid, 256, 0, a.stream>>>(a.S, a.X, a.Y, a.Z, a.lev_off, a.lo, a.hi, a.coef, a.AX, a.AY, a.AZ,
                                                         a.POT, a.VX, a.VY, a.VZ, a.assign);
}

void expamd_sph_thin_acc_gen(const SphThinAccArgs &a)
{
  if (a.n == 0) return;
  const size_t nrows = (size_t)a.S.nrows, lsn = (size_t)(a.S.lmax + 1) * a.S.nmax;
  const int tile0 = (int)EXPAMD_EXPT("EXP_AMD_THIN_TILE", 64);
  int tile = tile0 < 4 ? 4 : tile0 > SPH_TILE_MAX ? SPH_TILE_MAX : tile0;
  auto need = [&](int t) { return ((((size_t)t * (nrows | 1) + 1) & ~(size_t)1) + (size_t)t * lsn) * sizeof(double); };
  while (tile > 4 && need(tile) > 120 * 1024) tile >>= 1;
  // (ncoef <= SPH_TILE_KMAX * 256: the caller checks, sph.hip)
  size_t grid = cdiv(a.n, (size_t)tile);
  if (grid > 4096) grid = 4096;
  static const bool big = expamd_big_lds(reinterpret_cast<const void *>(&k_sph_acc_tile), "k_sph_acc_tile");
  (void)big;
  k_sph_acc_tile<<<(unsigned)grid, 256, need(tile), a.stream>>>(a.S, a.X, a.Y, a.Z, a.M, a.lev_off, a.lo, a.hi, a.wscale, a.part,
                                                              a.used, tile);
}
