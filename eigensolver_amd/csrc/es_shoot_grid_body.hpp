// Body of the fp64 grid kernels (included by es_shoot.hip inside shoot_grid_kernel with ROWS = 1 and inside
// shoot_grid_kernel_r2 with ROWS = 2; template parameters FAM, PTS, MAXT, TRACK, WPE and the kernel arguments are in scope).
// One text for both so that a point's arithmetic is the same; two kernels rather than one device function because the
// register allocation of the sign-tracking shapes came out 30 - 50 registers worse through an inlined function.
  constexpr int NE = FamTraits<FAM>::NE;
  // even row stride and a 16-byte aligned table: the entries of nodes 2j, 2j+1 of every row are one aligned 16-byte pair,
  // read by ds_read_b128 at an immediate offset from ONE address register (with the odd stride every other row needed
  // ds_read2_b64 from its own base register: 7 address moves per loop iteration, 2 % of its VALU instructions)
  constexpr int LSTRIDE = 2 * CH + 2;
  // two RK4 steps per loop iteration where the registers allow it (two coefficients per point, 4 points per lane)
  constexpr bool PAIR = (FAM == FAM_CYL0) && !TRACK;
  // register-capped instantiations (WPE != 0) park the exterior results in LDS during the march instead of letting
  // the compiler spill them to scratch (HBM): 4 doubles per point, lane-contiguous (conflict-free)
  constexpr bool STASH = (WPE != 0);
  __shared__ double xstash[STASH ? 4 * PTS * MAXT : 1];
  __shared__ __attribute__((aligned(16))) double lds_all[ROWS * NE * LSTRIDE];
  const int TW = blockDim.x;                           // threads of the workgroup
  const int T = TW / ROWS;                             // lanes of one k-row
  const int sub = (ROWS == 2 && (int)threadIdx.x >= T) ? 1 : 0;       // which of the workgroup's rows this wave marches
  const int tl = (int)threadIdx.x - sub * T;           // lane within the row
  const int tb = (ROWS == 2) ? sub * (NE * LSTRIDE) : 0;   // offset of the row's node table in lds_all (wave-uniform)
  const int nsteps = P.n_nodes - 1;
  const double h = P.h, h2 = 0.5 * P.h, h6 = P.h / 6.0, h3 = P.h / 3.0;

  // tile = (k-row, omega-segment of T*PTS points): rows wider than one segment are split across workgroups, so the
  // number of workgroups is nk * nseg (narrow k-tiles of a multi-GPU run still fill the chip, and the tail of the
  // launch is one segment long instead of one row)
  const int span = T * PTS;
  const int ncols = opts.cols ? opts.cols[0] : nw;     // columns to evaluate (workgroup-uniform)
  const int col_base = (opts.part == 2) ? (ncols / opts.main_span) * opts.main_span : 0;
  const int nseg = (opts.part == 2) ? opts.main_span / span : (nw + span - 1) / span;
  const int ngroups = (nk + ROWS - 1) / ROWS;          // groups of ROWS consecutive k-rows
  const long ntiles = (long)ngroups * nseg;
  for (long tile = es_tile_index(), once = 1; once && tile < ntiles; once = 0) {
    // segment-major order: consecutive workgroup ids (dealt round-robin to the 8 XCDs) are consecutive k-rows of one
    // omega-segment.  With the row-major order tile = row * nseg + segment and nseg = 4, segment s of every row went to
    // XCDs s and s + 4: segments whose points are dead (continuum, leaky) or absent (compacted launch) idled two XCDs
    // while the other six carried the launch
    const int seg = (int)(tile / ngroups);
    const int grp = (int)(tile - (long)seg * ngroups);
    const bool row_ok = (ROWS == 1) || grp * ROWS + sub < nk;   // an odd number of rows leaves the last group's second half empty
    const int row = (ROWS == 1) ? grp : (row_ok ? grp * ROWS + sub : nk - 1);
    const int w0 = col_base + seg * span;
    if (w0 >= ncols) continue;                         // segment beyond the live columns
    if (opts.part == 1 && w0 + span > ncols) continue; // partly filled segment: left to the remainder launch
    const double k = kv[row];
    const KScal s = make_kscal(P, k);
    {
      double w[PTS], zp[PTS], zq[PTS];
      Coef B0[PTS], B1[PTS];
      SignTrack trk[PTS];
      bool inr[PTS];
      int iwp[PTS];
      // exterior closed form first: here nothing of the march is live, so the ~100 VGPRs of the Bessel code overlap with
      // nothing and only its results are carried through the march (no call frame).
      // a wave none of whose points has an evanescent exterior (leaky / non-finite: D is NaN whatever the march gives)
      // only takes part in the LDS staging and the barriers
      ExteriorLite X[PTS];
      bool lane_live = false;
      if (STASH) {
        // ONE copy of the exterior code, executed PTS times (not unrolled): the results go to LDS at once, the frequency
        // is formed again for the march below (same operations, same value) -- so the register cap of the shape costs no
        // spill in the Bessel code and the kernel carries one copy of it instead of PTS
#pragma unroll 1
        for (int p = 0; p < PTS; ++p) {
          const int ic = w0 + p * T + tl;
          const bool in = row_ok && ic < ncols;
          const int iw = in ? (opts.cols ? opts.cols[1 + ic] : ic) : 0;
          const double wp = in ? pick_w(wv, w_mode, k, row, nw, iw) : 1.0;
          const ExteriorLite Xp = exterior_lite(P, k, wp, wp);
          // with ES_EVAL_SKIP_CONTINUUM a point inside a continuum band (known before the march) is not worth a march
          const bool dead = opts.skip && !TRACK && band_crossed(P, k, wp);
          lane_live = lane_live || (in && Xp.status == ES_PT_OK && !dead);
          double* xs = xstash + (size_t)(4 * p) * MAXT + threadIdx.x;   // lane-contiguous: conflict-free
          xs[0] = Xp.outer; xs[MAXT] = Xp.yb; xs[2 * MAXT] = Xp.Oe; xs[3 * MAXT] = (double)Xp.status;
        }
      }
#pragma unroll
      for (int p = 0; p < PTS; ++p) {
        const int ic = w0 + p * T + tl;
        inr[p] = row_ok && ic < ncols;
        iwp[p] = inr[p] ? (opts.cols ? opts.cols[1 + ic] : ic) : 0;
        w[p] = inr[p] ? pick_w(wv, w_mode, k, row, nw, iwp[p]) : 1.0;
      }
      if (!STASH) {
#pragma unroll
        for (int p = 0; p < PTS; ++p) {
          X[p] = exterior_lite(P, k, w[p], w[p]);
          const bool dead = opts.skip && !TRACK && band_crossed(P, k, w[p]);
          lane_live = lane_live || (inr[p] && X[p].status == ES_PT_OK && !dead);
        }
      }
      const bool wave_live = __any(lane_live);
      // a workgroup without any evanescent point (a whole omega-segment of leaky / singular points) skips the march
      const bool wg_live = __syncthreads_or(wave_live ? 1 : 0) != 0;
#pragma unroll
      for (int p = 0; p < PTS; ++p) { zp[p] = 0.0; zq[p] = 0.0; }
      // adjoint march: chunks from the far end of the interior back to the boundary
      const int nchunks = wg_live ? (nsteps + CH - 1) / CH : 0;
      for (int c = nchunks - 1; c >= 0; --c) {
        const int c0 = c * CH;
        const int nst = (nsteps - c0 < CH) ? (nsteps - c0) : CH;
        __syncthreads();                               // previous chunk fully consumed
        // the entry of the chunk's far node (index 2 nst) is only read by the chunk that starts the march (its coefficients
        // come over in B0 from the chunk before otherwise): 2 nst entries = ONE pass of 256 threads for a full chunk, where
        // 2 nst + 1 took a second pass for a single entry
        const int nstage = 2 * nst + ES_FAR_NODE(c, nchunks);
        for (int i = threadIdx.x; i < ROWS * nstage; i += TW) {
          double b[FamTraits<FAM>::NB], e[NE];
          const int r = (ROWS == 2 && i >= nstage) ? 1 : 0;                // table of the workgroup's first / second row
          const int node = i - r * nstage;
          load_base<FAM>(P, 2 * c0 + node, b);
          if (ROWS == 1) {
            make_entry<FAM, fam_scaled<FAM>()>(b, s, e);
          } else {
            const int rr = grp * ROWS + r;
            const KScal sr = make_kscal(P, kv[rr < nk ? rr : nk - 1]);
            make_entry<FAM, fam_scaled<FAM>()>(b, sr, e);
          }
#pragma unroll
          for (int f = 0; f < NE; ++f) lds_all[r * (NE * LSTRIDE) + f * LSTRIDE + node] = e[f];
        }
        __syncthreads();
        if (!wave_live) continue;
        if (c == nchunks - 1) {                        // last node: start vector of the march
          double eL[NE];
#pragma unroll
          for (int f = 0; f < NE; ++f) eL[f] = lds_all[tb + f * LSTRIDE + 2 * nst];
#pragma unroll
          for (int p = 0; p < PTS; ++p) {
            coefficients<FAM, TRACK>(eL, P, s, w[p], B0[p], trk[p]);
            adjoint_start(P, B0[p], zp[p], zq[p]);
          }
        }
        // one RK4 step of all PTS points: coefficients of node 2j+1 / 2j from LDS (broadcast reads), start
        // coefficients BIN, end coefficients written to BOUT (the next step's start)
#define ES_MARCH_STEP(J, BIN, BOUT)                                                         \
        {                                                                                   \
          double em[NE], e1[NE];                                                            \
          _Pragma("unroll") for (int f = 0; f < NE; ++f) {                                  \
            em[f] = lds_all[tb + f * LSTRIDE + 2 * (J) + 1];                                         \
            e1[f] = lds_all[tb + f * LSTRIDE + 2 * (J)];                                             \
          }                                                                                 \
          _Pragma("unroll") for (int p = 0; p < PTS; ++p) {                                 \
            Coef Bm;                                                                        \
            coefficients2<FAM, TRACK>(em, e1, P, s, w[p], Bm, BOUT[p], trk[p]);             \
            adjoint_step<FAM>(zp[p], zq[p], BIN[p], Bm, BOUT[p], h, h2, h6, h3);                             \
          }                                                                                 \
        }
        // two RK4 steps of all PTS points with ONE division per point (coefficients4: families with fam_rcp4()): nodes
        // 2J+1, 2J (step J) and 2J-1, 2J-2 (step J-1); start coefficients BIN, end coefficients of step J-1 to BOUT
#define ES_MARCH_PAIR(J, BIN, BOUT)                                                         \
        {                                                                                   \
          double em[NE], e1[NE], em2[NE], e12[NE];                                          \
          _Pragma("unroll") for (int f = 0; f < NE; ++f) {                                  \
            em[f] = lds_all[tb + f * LSTRIDE + 2 * (J) + 1];                                         \
            e1[f] = lds_all[tb + f * LSTRIDE + 2 * (J)];                                             \
            em2[f] = lds_all[tb + f * LSTRIDE + 2 * (J) - 1];                                        \
            e12[f] = lds_all[tb + f * LSTRIDE + 2 * (J) - 2];                                        \
          }                                                                                 \
          _Pragma("unroll") for (int p = 0; p < PTS; ++p) {                                 \
            Coef Bm, Bj, Bm2;                                                               \
            coefficients4<FAM, TRACK>(em, e1, em2, e12, P, s, w[p], Bm, Bj, Bm2, BOUT[p], trk[p]); \
            adjoint_step<FAM>(zp[p], zq[p], BIN[p], Bm, Bj, h, h2, h6, h3);                 \
            adjoint_step<FAM>(zp[p], zq[p], Bj, Bm2, BOUT[p], h, h2, h6, h3);               \
          }                                                                                 \
        }
        int j = nst - 1;
        if (fam_rcp4<FAM>()) {
          // step j with step j - 1 for every odd j (the pairing every fp64 march of the family uses: coefficients4); an
          // even top step alone; the pairs two per iteration with the roles of B0 / B1 swapped (no coefficient copies)
          if (nst & 1) {
            ES_MARCH_STEP(j, B0, B1)
#pragma unroll
            for (int p = 0; p < PTS; ++p) B0[p] = B1[p];
            --j;
          }
          if (TRACK && FAM != FAM_CYL0) {
            // the sign-tracking fall-backs of the slab families (profiles whose continuum intervals do not overlap) keep
            // three or four watched terms per point: one pair per iteration (two would not fit 256 registers)
            for (; j >= 1; j -= 2) {
              ES_MARCH_PAIR(j, B0, B1)
#pragma unroll
              for (int p = 0; p < PTS; ++p) B0[p] = B1[p];
            }
          } else {
            if (((j + 1) >> 1) & 1) {                  // odd number of pairs: one ahead of the loop
              ES_MARCH_PAIR(j, B0, B1)
#pragma unroll
              for (int p = 0; p < PTS; ++p) B0[p] = B1[p];
              j -= 2;
            }
            for (; j >= 3; j -= 4) {
              ES_MARCH_PAIR(j, B0, B1)
              ES_MARCH_PAIR(j - 2, B1, B0)
            }
          }
        } else if (PAIR) {
          // steps in pairs with the roles of B0 / B1 swapped, so that no coefficient is copied between iterations
          if (nst & 1) {
            ES_MARCH_STEP(j, B0, B1)
#pragma unroll
            for (int p = 0; p < PTS; ++p) B0[p] = B1[p];
            --j;
          }
          for (; j >= 1; j -= 2) {
            ES_MARCH_STEP(j, B0, B1)
            ES_MARCH_STEP(j - 1, B1, B0)
          }
        } else {
          for (; j >= 0; --j) {
            ES_MARCH_STEP(j, B0, B1)
#pragma unroll
            for (int p = 0; p < PTS; ++p) B0[p] = B1[p];
          }
        }
#undef ES_MARCH_PAIR
#undef ES_MARCH_STEP
#pragma unroll
        for (int p = 0; p < PTS; ++p) adjoint_rescale<FAM>(zp[p], zq[p], nsteps - c0 - nst, nsteps - c0);
      }
      if (STASH) {
#pragma unroll
        for (int p = 0; p < PTS; ++p) {
          const double* xs = xstash + (size_t)(4 * p) * MAXT + threadIdx.x;
          X[p].outer = xs[0]; X[p].yb = xs[MAXT]; X[p].Oe = xs[2 * MAXT]; X[p].status = (int)xs[3 * MAXT];
        }
      }
      // boundary: exterior closed form + far-end condition + mismatch
      double bf[FamTraits<FAM>::NB], ef[NE];
      load_base<FAM>(P, 0, bf);
      make_entry<FAM>(bf, s, ef);
#pragma unroll
      for (int p = 0; p < PTS; ++p) {
        if (!inr[p]) continue;
        const int iw = iwp[p];
        const Mismatch M = boundary_algebra<FAM>(P, s, w[p], X[p], zp[p], zq[p], ef);
        double D, rel; uint8_t st;
        const bool crossed = TRACK ? trk[p].crossed() : band_crossed(P, k, w[p]);
        finish_point(P, M, X[p], crossed, D, rel, st);
        // a band point of ES_EVAL_SKIP_CONTINUUM may not have been marched at all (a wave or workgroup of dead points keeps
        // z = 0, the boundary algebra gives 0/0 and finish_point says NONFINITE): its status is CONTINUUM, as the header
        // documents (the fp32 screening kernel does the same)
        if (opts.skip && !TRACK && crossed && X[p].status == ES_PT_OK) st = ES_PT_CONTINUUM;
        if (opts.skip && st == ES_PT_CONTINUUM) { D = NAN; rel = NAN; }
        const size_t o = (size_t)row * nw + iw;
        Dout[o] = D;
        stout[o] = st;
        if (relout) relout[o] = rel;
      }
    }
  }
