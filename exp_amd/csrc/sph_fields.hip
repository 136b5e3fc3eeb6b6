// Field evaluation at arbitrary points for the spherical basis: the pyEXP surface
// Spherical::sph_eval / cyl_eval / crt_eval behind getFields (expui/BiorthBasis.cc:711-816,
// :930-958).  Not a throughput path (grids of <= 1e6 points): one lane per point, run-time loops
// over (l, m), per-lane gathers from the projected tables
//   G [i][row] = sum_n ef_l(n,i)/sqrt(ev_l[n]) c[row][n]     (potential / force, sph_project)
//   Gd[i][row] = sum_n ef_l(n,i)*sqrt(ev_l[n]) c[row][n]     (density, SLGridMP2.cc:913-950)
// so a point costs O(L^2) like the n-body force pass, not O(L^2 nmax).
#include "sph_force.h"

// Gd[i][row] = sum_n E[i][l][n] ev[l][n] c[row][n]
__global__ void __launch_bounds__(256)
k_sph_project_dens(SphDev S, const double *__restrict__ ev, const double *__restrict__ coef,
                   double *__restrict__ Gd)
{
  const int i = blockIdx.x;
  const int stride = (S.lmax + 1) * S.nmax;
  for (int row = threadIdx.x; row < S.nrows; row += 256) {
    int l = 0;
    while ((l + 1) * (l + 1) <= row) l++;
    const double *e = S.E + (size_t)i * stride + l * S.nmax;
    const double *v = ev + (size_t)l * S.nmax;
    const double *c = coef + (size_t)row * S.nmax;
    double s = 0.0;
    for (int n = 0; n < S.nmax; n++) s = fma(e[n] * v[n], c[n], s);
    Gd[(size_t)i * S.nrows + row] = s;
  }
}

__global__ void __launch_bounds__(256)
k_sph_fields(SphDev S, const double *__restrict__ G, const double *__restrict__ Gd,
             const double *__restrict__ d0, size_t n, const double *__restrict__ c1,
             const double *__restrict__ c2, const double *__restrict__ c3, int coord,
             double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double *lc = S.lc;
  const int L = S.lmax;
  // ---- coordinates (cyl_eval :930-941, crt_eval :946-958) ----
  double r, costh, phi, R = 0.0, x = 0.0, y = 0.0, z = 0.0;
  if (coord == 0) { r = c1[i]; costh = c2[i]; phi = c3[i]; }
  else {
    if (coord == 1) { R = c1[i]; z = c2[i]; phi = c3[i]; }
    else {
      x = c1[i]; y = c2[i]; z = c3[i];
      R = sqrt(sq_sum2_lit(x, y)) + 1.0e-18;           // (every product rounded on its own, as the reference's compiler
      phi = atan2(y, x);                                //  forms them: near the polar axis an ulp of r decides cos(theta),
    }                                                   //  see sq_sum2_lit, sph_kernels.h)
    r = sqrt(sq_sum2_lit(R, z)) + 1.0e-18;
    costh = z / r;
  }
  // ---- radial tables: get_dens / get_pot / get_force (exputil/SLGridMP2.cc:872-989) ----
  const double xi = sph_r_to_xi(S, r / S.scale);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) / S.dxi;
  const double x2 = (xi - S.xi[idx]) / S.dxi;
  const double P0 = x1 * S.p0[idx] + x2 * S.p0[idx + 1];
  const double D0 = x1 * d0[idx] + x2 * d0[idx + 1];
  const int j = idx < 1 ? 1 : idx;
  const double pf = (xi - S.xi[j]) / S.dxi;
  const double ffac = sph_d_xi_to_r(S, xi) / S.dxi;
  const double wa = (pf - 0.5) * S.p0[j - 1], wb = -2.0 * pf * S.p0[j], wc = (pf + 0.5) * S.p0[j + 1];
  const bool lit = pf < S.lit_lo || pf > S.lit_hi;
  const double *g0 = G + (size_t)idx * S.nrows, *g1 = g0 + S.nrows;
  const double *ga = G + (size_t)(j - 1) * S.nrows, *gb = ga + S.nrows, *gc = gb + S.nrows;
  const double *q0 = Gd + (size_t)idx * S.nrows, *q1 = q0 + S.nrows;
  // ---- angular part: normalised Legendre functions and their x-derivative ----
  double xc = costh;                                   // pole clamp of the derivative (:1109-1112)
  if (1.0 - fabs(xc) < MINEPS) xc = (xc > 0) ? 1.0 - MINEPS : -(1.0 - MINEPS);
  const double dfac = 1.0 / sq_add_lit(-1.0, xc);
  const double somx2 = sqrt((1.0 - costh) * (1.0 + costh));
  const bool polar = 1.0 - fabs(costh) < SPH_POLAR_FAC;         // the m = 0 derivative by the reference's own recurrence
  double lp1 = 0.0, lp2 = 0.0;                                  // (leg0_lit_step, sph_kernels.h)
  double den0 = 0.0, pot0 = 0.0, potr = 0.0, den1 = 0.0, pot1 = 0.0, pott = 0.0, potp = 0.0;
  double pmm = 0.0;
  for (int m = 0; m <= L; m++) {
    pmm = (m == 0) ? lc[3] : pmm * (lc[((size_t)m * (L + 1) + m) * 4 + 3] * somx2);
    const bool m_on = !((S.M0_only && m) || (S.EVEN_M && (m & 1)));
    const double cosm = cos(phi * m), sinm = sin(phi * m);     // direct, as sph_eval does
    double pl2 = 0.0, pl1 = 0.0;
    for (int l = m; l <= L; l++) {
      const double *q = lc + ((size_t)l * (L + 1) + m) * 4;
      double plm, dplm;
      if (l == m) { plm = pmm; dplm = dfac * ((xc * l) * plm); }
      else {
        plm = (l == m + 1) ? q[0] * (costh * pl1) : q[0] * (costh * pl1) - q[1] * pl2;
        dplm = dfac * ((xc * l) * plm - q[2] * pl1);
      }
      if (m == 0 && polar) {
        double ql;
        leg0_lit_step(l, costh, xc, lp1, lp2, ql);
        dplm = dfac * (ql * (plm / lp1));
      }
      pl2 = pl1;
      pl1 = plm;
      bool on = m_on;
      if (l == 0 && S.NO_L0) on = false;
      if (l == 1 && S.NO_L1) on = false;
      if (l > 0 && S.EVEN_L && (l & 1)) on = false;
      if (!on) continue;
      // coefficient row, with the reference's EVEN_M quirk (skipped odd m do not advance moffset)
      int row = l * l + (m ? 2 * m - 1 : 0);
      if (S.EVEN_M && m > 0) row = l * l + (m - 1);
      const double sumP0 = P0 * (x1 * g0[row] + x2 * g1[row]);
      const double sumR0 = D0 * (x1 * q0[row] + x2 * q1[row]);
      // (far outside the table -- the logarithmic map only -- the radial derivative is formed term by term as the
      // reference does: sph_dp_lit_row, sph_kernels.h)
      const double sumD0 = ffac * (lit ? sph_dp_lit_row(S, row, l, pf) : wa * ga[row] + wb * gb[row] + wc * gc[row]);
      if (m == 0) {
        if (l == 0) { den0 = plm * sumR0; pot0 = plm * sumP0; potr += plm * sumD0; }
        else {
          den1 += plm * sumR0; pot1 += plm * sumP0; potr += plm * sumD0; pott += dplm * sumP0;
        }
      } else {
        const double sumP1 = P0 * (x1 * g0[row + 1] + x2 * g1[row + 1]);
        const double sumR1 = D0 * (x1 * q0[row + 1] + x2 * q1[row + 1]);
        const double sumD1 = ffac * (lit ? sph_dp_lit_row(S, row + 1, l, pf)
                                         : wa * ga[row + 1] + wb * gb[row + 1] + wc * gc[row + 1]);
        den1 += plm * (sumR0 * cosm + sumR1 * sinm);
        pot1 += plm * (sumP0 * cosm + sumP1 * sinm);
        potr += plm * (sumD0 * cosm + sumD1 * sinm);
        pott += dplm * (sumP0 * cosm + sumP1 * sinm);
        potp += plm * (-sumP0 * sinm + sumP1 * cosm) * m;
      }
    }
  }
  double cc = costh * costh;
  asm volatile("" : "+v"(cc));                          // (rounded before the subtraction, expui/BiorthBasis.cc:726)
  const double sinth = sqrt(fabs(1.0 - cc));
  const double densfac = 1.0 / (S.scale * S.scale * S.scale) * 0.25 / M_PI;
  const double potlfac = 1.0 / S.scale;
  double v[9] = {den0 * densfac, den1 * densfac, (den0 + den1) * densfac,
                 pot0 * potlfac, pot1 * potlfac, (pot0 + pot1) * potlfac,
                 potr * (-potlfac) / S.scale, pott * (-potlfac) / r, potp * (-potlfac) / (r * sinth)};
  double *o = out + 9 * i;
  if (coord != 0) {
    const double sth = R / r;
    const double potR = v[6] * sth - v[7] * costh * R / r;
    const double potz = v[6] * costh + v[7] * sth * R / r;
    if (coord == 1) { v[6] = potR; v[7] = potz; }
    else {
      const double fx = potR * x / R - v[8] * y / R;
      const double fy = potR * y / R + v[8] * x / R;
      v[6] = fx; v[7] = fy; v[8] = potz;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = v[k];
}

extern "C" int exp_amd_sph_set_density(exp_amd_force *fb, const double *d0)
{
  expamd_mutated();            // (drops a captured graph of fused steps: exp_amd_step_kdk_n)
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !d0) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "sph_set_density: not a spherical force / NULL");
  exp_amd_ctx *ctx = f->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (f->d_d0.alloc(f->cfg.numr) != hipSuccess || f->d_Gd.alloc((size_t)f->cfg.numr * f->dev.nrows) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_set_density: hipMalloc failed");
  HIP_TRY(ctx, hipMemcpyAsync(f->d_d0.p, d0, f->cfg.numr * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EXP_AMD_OK;
}

extern "C" int exp_amd_sph_fields(exp_amd_force *fb, size_t n, const double *c1, const double *c2,
                                  const double *c3, int coord, double *out)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "sph_fields: not a spherical force");
  exp_amd_ctx *ctx = f->ctx;
  if (n == 0) return EXP_AMD_OK;
  if (!c1 || !c2 || !c3 || !out || coord < 0 || coord > 2)
    return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_fields: bad argument");
  if (!f->d_d0.p)
    return expamd_fail(ctx, EXP_AMD_ERR_STATE, "sph_fields: call exp_amd_sph_set_density first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = sph_project(f);
  if (rc) return rc;
  DevBuf<double> d_in, d_out;
  if (d_in.alloc(3 * n) != hipSuccess || d_out.alloc(9 * n) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_fields: hipMalloc failed");
  const double *src[3] = {c1, c2, c3};
  for (int k = 0; k < 3; k++)
    HIP_TRY(ctx, hipMemcpyAsync(d_in.p + (size_t)k * n, src[k], n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  k_sph_project_dens<<<f->cfg.numr, 256, 0, ctx->stream>>>(f->dev, f->d_ev.p, f->d_coef.p, f->d_Gd.p);
  k_sph_fields<<<cdiv(n, 256), 256, 0, ctx->stream>>>(f->dev, f->d_G.p, f->d_Gd.p, f->d_d0.p, n, d_in.p,
                                                      d_in.p + n, d_in.p + 2 * n, coord, d_out.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, 9 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_in.release();
  d_out.release();
  return EXP_AMD_OK;
}

// ---- the basis functions themselves on a radial grid (pyEXP getBasis) ---------------------------------
// SphericalSL::getBasis (expui/BiorthBasis.cc:960-993) tabulates SLGridSph::get_pot / get_dens /
// get_force (exputil/SLGridMP2.cc:872-989) of every (l, n) -- at r itself, not r/scale -- and returns
// the force with its sign changed.  One lane per radius, blockIdx.y = l * nmax + n.
__global__ void __launch_bounds__(256)
k_sph_basis(SphDev S, const double *__restrict__ ev, const double *__restrict__ d0, size_t n,
            const double *__restrict__ r, double *__restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ln = blockIdx.y;                                  // l * nmax + n
  const int stride = (S.lmax + 1) * S.nmax;
  const double xi = sph_r_to_xi(S, r[i]);
  const int idx = sph_cell(S, xi);
  const double x1 = (S.xi[idx + 1] - xi) / S.dxi;
  const double x2 = (xi - S.xi[idx]) / S.dxi;
  const double u = x1 * S.E[(size_t)idx * stride + ln] + x2 * S.E[(size_t)(idx + 1) * stride + ln];
  const int j = idx < 1 ? 1 : idx;
  const double pf = (xi - S.xi[j]) / S.dxi;
  const double ffac = sph_d_xi_to_r(S, xi) / S.dxi;
  const double frc = ffac * ((pf - 0.5) * S.E[(size_t)(j - 1) * stride + ln] * S.p0[j - 1]
                             - 2.0 * pf * S.E[(size_t)j * stride + ln] * S.p0[j]
                             + (pf + 0.5) * S.E[(size_t)(j + 1) * stride + ln] * S.p0[j + 1]);
  const size_t plane = (size_t)stride * n;
  double *o = out + (size_t)ln * n + i;
  o[0] = u * (x1 * S.p0[idx] + x2 * S.p0[idx + 1]);                       // potential
  o[plane] = u * ev[ln] * (x1 * d0[idx] + x2 * d0[idx + 1]);              // density: ef sqrt(ev) d0 = E ev d0
  o[2 * plane] = -frc;                                                    // radial force
}

extern "C" int exp_amd_sph_basis(exp_amd_force *fb, size_t n, const double *r, double *out)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f) return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "sph_basis: not a spherical force");
  exp_amd_ctx *ctx = f->ctx;
  if (n == 0) return EXP_AMD_OK;
  if (!r || !out) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_basis: NULL argument");
  if (!f->d_d0.p) return expamd_fail(ctx, EXP_AMD_ERR_STATE, "sph_basis: call exp_amd_sph_set_density first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t ln = (size_t)(f->cfg.lmax + 1) * f->cfg.nmax;
  DevBuf<double> d_r, d_out;
  if (d_r.alloc(n) != hipSuccess || d_out.alloc(3 * ln * n) != hipSuccess)
    return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_basis: hipMalloc failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_r.p, r, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  k_sph_basis<<<dim3(cdiv(n, 256), (unsigned)ln), 256, 0, ctx->stream>>>(f->dev, f->d_ev.p, f->d_d0.p, n, d_r.p, d_out.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, 3 * ln * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_r.release();
  d_out.release();
  return EXP_AMD_OK;
}

// ---- mass inside the expansion window (pyEXP getMass) --------------------------------------------------
// Spherical::accumulate adds a particle's mass to totalMass when rmin <= r <= rmax, r = sqrt(r^2) + dsmall
// (expui/BiorthBasis.cc:596-607); BiorthBasis::getMass returns it (expui/BiorthBasis.H:189).
__global__ void __launch_bounds__(256)
k_sph_window_mass(SphDev S, size_t n, const double *__restrict__ x, const double *__restrict__ y,
                  const double *__restrict__ z, const double *__restrict__ m, double *__restrict__ out)
{
  __shared__ double part[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double xx = x[i] - S.cx, yy = y[i] - S.cy, zz = z[i] - S.cz;
    const double r = sqrt(xx * xx + yy * yy + zz * zz) + S.dsmall;
    if (!(r < S.rmin || r > S.rmax)) s += S.umass != 0.0 ? S.umass : m[i];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

extern "C" int exp_amd_sph_window_mass(exp_amd_force *fb, exp_amd_comp *c, double *mass)
{
  SphForce *f = dynamic_cast<SphForce *>(fb);
  if (!f || !c || !mass)
    return expamd_fail(fb ? fb->ctx : nullptr, EXP_AMD_ERR_ARG, "sph_window_mass: not a spherical force / NULL");
  { int rc_ = expamd_comp_densify(c); if (rc_) return rc_; }      // (an appended store: made an ordinary one first)
  exp_amd_ctx *ctx = f->ctx;
  if (c->ctx != ctx) return expamd_fail(ctx, EXP_AMD_ERR_ARG, "sph_window_mass: component of another context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  *mass = 0.0;                                  // (positions and masses only: a pending half-kick does not matter)
  if (c->n == 0) return EXP_AMD_OK;
  DevBuf<double> d_s;
  if (d_s.alloc(1) != hipSuccess) return expamd_fail(ctx, EXP_AMD_ERR_HIP, "sph_window_mass: hipMalloc failed");
  HIP_TRY(ctx, hipMemsetAsync(d_s.p, 0, sizeof(double), ctx->stream));
  SphDev S = f->dev;
  S.cx = c->center[0]; S.cy = c->center[1]; S.cz = c->center[2];
  S.umass = c->uniform_mass ? c->mass_value : 0.0;
  const unsigned nb = (unsigned)(cdiv(c->n, 256) < 2048 ? cdiv(c->n, 256) : 2048);
  k_sph_window_mass<<<nb, 256, 0, ctx->stream>>>(S, c->n, c->a(A_X), c->a(A_Y), c->a(A_Z), c->a(A_M), d_s.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(mass, d_s.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  d_s.release();
  return EXP_AMD_OK;
}
