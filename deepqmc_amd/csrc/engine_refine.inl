// engine_refine.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// float64 refinement of a float32 context: the twin, gather -> float64 pass -> write-back, self-calibration of the score
// threshold, the whole-batch float64 mode with hysteresis.

  int ensure_twin() {
    if (twin) return DQMC_OK;
    if (no_twin) return fail(DQMC_E_UNSUPPORTED, "the float64 twin of this context is switched off (option no_twin)");
    auto* t = new Engine<double>();
    t->st = st; t->device = device;
    dqmc_system s2 = sys;
    s2.dtype = 1;
    int rc = t->init(&s2, charges_h.data(), bufs.data(), (int)bufs.size(), ops.data(), (int)ops.size(), w64_h.data(), w64_h.size(),
                     h_itable.data(), h_itable.size());
    if (!rc && (!ecp_loc_h.empty() || !ecp_nl_h.empty()))      // the twin carries the whole ECP: its local energies include V_nl
      rc = t->set_ecp(ecp_loc_nt_h, ecp_loc_h.empty() ? nullptr : ecp_loc_h.data(), ecp_nl_L_h, ecp_nl_nt_h, ecp_nl_h.empty() ? nullptr : ecp_nl_h.data());
    if (!rc && !ph_mask_h.empty()) rc = t->set_ph(ph_grid, ph_rmax, ph_loc_h.data(), ph_l2_h.data(), ph_mask_h.data());
    for (size_t k = 0; k < twin_opts.size() && !rc; ++k) rc = t->option(twin_opts[k].first.c_str(), twin_opts[k].second);
    if (rc) { delete t; return rc; }
    t->ws_budget = twin_full_budget ? ws_budget : ws_budget / 2;
    t->timing = timing;
    twin = t;
    return DQMC_OK;
  }
  // float64 results of the n walkers listed in d_flag[1..n] replace the float32 ones.  d_count != nullptr: the list is
  // holds fewer than n entries -- the count is read on the device, n is the (padded) size of this pass (kernels_mcmc.hip:
  // k_refine_gather).
  int refine_listed(const real* r, const real* R, int B, int n, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign,
                    const int32_t* d_count = nullptr, const int32_t* d_list = nullptr, int n_scatter = -1, bool use_score = false) {
    if (!d_list) d_list = d_flag + 1;
    if (n_scatter < 0) n_scatter = n;
    const int n3 = 3 * N, nR3 = 3 * sys.n_nuc;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_r = 0, o_R = o_r + al(sizeof(double) * (size_t)n * n3), o_e = o_R + al(sizeof(double) * nR3),
                 o_s = o_e + al(sizeof(double) * n), o_g = o_s + al(sizeof(double) * 6 * (size_t)n),
                 o_l = o_g + al(sizeof(double) * (size_t)n * n3), o_sg = o_l + al(sizeof(double) * n),
                 tot = o_sg + al(sizeof(int32_t) * n);
    if (tot > ref_bytes) {
      if (d_ref) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ref)); d_ref = nullptr; ref_bytes = 0; }
      HIP_TRY(hipMalloc((void**)&d_ref, tot));
      ref_bytes = tot;
    }
    double* r64 = (double*)(d_ref + o_r); double* R64 = (double*)(d_ref + o_R); double* e64 = (double*)(d_ref + o_e);
    double* s64 = (double*)(d_ref + o_s); double* g64 = (double*)(d_ref + o_g); double* l64 = (double*)(d_ref + o_l);
    int32_t* sg64 = (int32_t*)(d_ref + o_sg);
    t_begin("refine", 0);
    dqmc::launch_refine_gather(st, (const float*)r, (const float*)R, d_list, d_count, n, n3, nR3, r64, R64);
    t_end();
    twin->ph_skip = (e_loc == nullptr);       // psi_grad / Langevin: the plain gradient, no pseudo-Hamiltonian seeding
    // a Hamiltonian with a non-local ECP: the twin runs the quadrature of its walkers in float64 with the rotation angles
    // of the walkers they stand for (its psi ratios carry the float64 value path's accuracy: float32 ratios alone put
    // ~1e-4 relative on E_loc of a 30-electron Psiformer)
    twin->ecp_skip_nl = (e_loc == nullptr) || ecp_defer;      // (deferred: ecp_mixed adds V_nl to every walker afterwards)
    static_cast<Engine<double>*>(twin)->ecp_seed = ecp_seed;
    static_cast<Engine<double>*>(twin)->ecp_phi = ecp_phi;
    static_cast<Engine<double>*>(twin)->ecp_phi_f32 = true;
    static_cast<Engine<double>*>(twin)->ecp_idx = d_list;
    const int rc = twin->local_energy(r64, R64, n, e64, s64, g64, l64, sg64);
    if (twin->last_chunks[0] > last_chunks[1]) last_chunks[1] = twin->last_chunks[0];
    twin->ph_skip = false;
    twin->ecp_skip_nl = false;
    if (rc) return rc;
    t_begin("refine", 0);
    // (entries past n_scatter -- the calibration sample of a probe call -- are evaluated, read by the host, not written back)
    ref_e64 = e64;
    if (use_score) {
      // probe call: the new threshold needs the float64 energies of the sample on the host first; then only the walkers
      // above it are written back (d_score / refine_thresh at that moment)
      std::vector<double> e_h((size_t)n);
      HIP_TRY(hipMemcpyAsync(e_h.data(), e64, sizeof(double) * e_h.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      probe_sample_e = e_h;
      if (probe_rethreshold) probe_rethreshold();
      n_scatter = n;                       // every evaluated walker above the NEW threshold is written back
    }
    dqmc::launch_refine_scatter(st, d_list, d_count, n, n_scatter, use_score ? d_score : nullptr, refine_thresh, n3, e64, s64, g64, l64, sg64,
                                (float*)e_loc, (float*)stats, (long)B, (float*)grad, (float*)logpsi, sign);
    t_end();
    if (!d_count && !use_score) last_refined += n_scatter;
    return DQMC_OK;
  }
  // ---- float64 tail of a float32 forward-Laplacian pass (engine.hip, above tail_f64) ----
  // May this (unchunked) pass of B walkers hand ops [k_tail, end) to the twin?  Creates the twin if need be.
  bool tail_ready(int B, int TP) {
    if constexpr (sizeof(real) == 4) {
      if (!tail_f64 || k_tail <= 0 || fused_dbg) return false;
      if (ensure_twin() != DQMC_OK) { g_err.clear(); return false; }      // (no float64 kernel set for this program: plain float32)
      auto* tw = static_cast<Engine<double>*>(twin);
      return (double)tw->ws_bytes_per_walker(TP, &tail_alloc) * (double)B <= (double)tw->ws_budget;
    }
    return false;
  }
  // what a captured pass with a tail holds besides this context's own workspace
  const void* tail_ws(int which) const {
    if constexpr (sizeof(real) == 4) {
      if (!tail_now || !twin) return nullptr;
      return which == 0 ? (const void*)static_cast<const Engine<double>*>(twin)->d_ws : (const void*)d_tail;
    }
    return nullptr;
  }
  // The head (ops [0, k_tail)) of this chunk has been enqueued on `st`: widen its live buffers into the twin's workspace,
  // let the twin execute the tail there, narrow the results into the caller's arrays.
  int run_tail(const real* r, const real* R, int B, real* logpsi, int32_t* sign, real* e_loc, real* stats, long stats_ld, real* grad,
               int b_offset) {
    if constexpr (sizeof(real) == 4) {
      auto* tw = static_cast<Engine<double>*>(twin);
      const int T = 3 * N + 2, TP = (T + 15) / 16 * 16, n3 = 3 * N, nR3 = 3 * sys.n_nuc;
      auto al = [](size_t x) { return (x + 255) / 256 * 256; };
      const size_t o_r = 0, o_R = o_r + al(sizeof(double) * (size_t)B * n3), o_e = o_R + al(sizeof(double) * nR3),
                   o_s = o_e + al(sizeof(double) * B), o_g = o_s + al(sizeof(double) * 6 * (size_t)B),
                   o_l = o_g + al(sizeof(double) * (size_t)B * n3), o_sg = o_l + al(sizeof(double) * B),
                   tot = o_sg + al(sizeof(int32_t) * B);
      if (tot > tail_bytes) {
        if (d_tail) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_tail)); d_tail = nullptr; tail_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&d_tail, tot));
        tail_bytes = tot;
      }
      double* r64 = (double*)(d_tail + o_r); double* R64 = (double*)(d_tail + o_R); double* e64 = (double*)(d_tail + o_e);
      double* s64 = (double*)(d_tail + o_s); double* g64 = (double*)(d_tail + o_g); double* l64 = (double*)(d_tail + o_l);
      int32_t* sg64 = (int32_t*)(d_tail + o_sg);
      const hipStream_t keep_st = tw->st;
      tw->st = st;                                     // (a captured pass runs on the capture stream)
      int rc = tw->plan(B, TP, &tail_alloc);
      if (rc) { tw->st = keep_st; return rc; }
      t_begin("tail", 0);
      dqmc::launch_widen(st, (const float*)r, r64, (long)B * n3);
      dqmc::launch_widen(st, (const float*)R, R64, (long)nR3);
      tw->src32_of.assign(bufs.size(), nullptr);
      for (int b : tail_in) {
        if (tail_direct[b]) tw->src32_of[b] = (const float*)bptr(b);       // read in place by the twin's linear kernel
        else dqmc::launch_widen(st, (const float*)bptr(b), tw->bptr(b), (long)B * bufs[b].rows * lanes_of(b, TP) * bufs[b].width);
      }
      t_end();
      // the twin's k_final flags and scores for THIS context (same walker numbering: b_offset + walker of the chunk)
      const bool k_flag = tw->flag_on; int32_t* const k_dflag = tw->d_flag; double* const k_dscore = tw->d_score;
      const double k_thr = tw->refine_thresh; double* const k_dthr = tw->d_thresh;
      tw->flag_on = flag_on; tw->d_flag = d_flag; tw->d_score = d_score; tw->refine_thresh = refine_thresh; tw->d_thresh = d_thresh;
      tw->ph_skip = (e_loc == nullptr);
      tw->tail_only = true;
      tw->in_tail = in_tail; tw->tail_alloc = tail_alloc;      // (the twin's own analysis of the same program gives the same sets)
      rc = tw->run_chunk(r64, R64, B, true, l64, sg64, e_loc ? e64 : nullptr, s64, (long)B, grad ? g64 : nullptr, b_offset);
      tw->tail_only = false;
      tw->src32_of.clear();
      tw->ph_skip = false;
      tw->flag_on = k_flag; tw->d_flag = k_dflag; tw->d_score = k_dscore; tw->refine_thresh = k_thr; tw->d_thresh = k_dthr;
      tw->st = keep_st;
      if (rc) return rc;
      t_begin("tail", 0);
      dqmc::launch_tail_narrow(st, B, n3, e64, s64, g64, l64, sg64, (float*)e_loc, e_loc ? (float*)stats : nullptr, stats_ld, (float*)grad,
                               (float*)logpsi, sign);
      t_end();
      HIP_TRY(hipGetLastError());
    }
    return DQMC_OK;
  }

  // error-predictor scores of the last float32 pass that flagged (host copy; walkers of that call, in order)
  int last_score_B = 0;
  int refine_scores(double* out, int n) override {
    if (sizeof(real) != 4 || !d_score || last_score_B < 1) return fail(DQMC_E_UNSUPPORTED, "no float32 pass with the error predictor has run on this context");
    if (n > last_score_B) return fail(DQMC_E_ARG, "more scores requested than the last flagged pass had walkers");
    HIP_TRY(hipMemcpyAsync(out, d_score, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }
  int upload_list(const std::vector<int32_t>& idx) {
    std::vector<int32_t> buf(idx.size() + 1);
    buf[0] = (int32_t)idx.size();
    std::copy(idx.begin(), idx.end(), buf.begin() + 1);
    HIP_TRY(hipMemcpyAsync(d_flag, buf.data(), sizeof(int32_t) * buf.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }
  int lap_refined(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    const int rc = lap_refined_ecp(r, R, B, e_loc, stats, grad, logpsi, sign);
    refine_info[0] = sizeof(real) == 8 ? 0 : refine; refine_info[1] = refine_thresh; refine_info[2] = probe_c; refine_info[3] = refine_all_calls;
    return rc;
  }
  int lap_refined_ecp(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    if constexpr (sizeof(real) == 4) {
      if (refine == 1 && ecp_mixed_on && ecp_n_nl > 0 && e_loc && !ecp_skip_nl) {
        // kinetic part first (float32 pass, flagged walkers re-run in float64 WITHOUT the quadrature), then V_nl of every
        // walker with the precision chosen per (nucleus, electron) pair
        ecp_defer = true;
        int rc = lap_refined_core(r, R, B, e_loc, stats, grad, logpsi, sign);
        ecp_defer = false;
        if (rc) return rc;
        return ecp_mixed((const float*)r, (const float*)R, B, (float*)e_loc, (float*)stats);
      }
    }
    return lap_refined_core(r, R, B, e_loc, stats, grad, logpsi, sign);
  }
  // Non-local ECP term of a float32 context with per-pair precision (kernels_ecp.hip: "mixed-precision quadrature"):
  // added to e_loc, stored in stats[3].
  int lap_refined_core(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    const int rc = lap_refined_core_(r, R, B, e_loc, stats, grad, logpsi, sign);
    refine_counters[3] += last_refined;
    return rc;
  }
  // more than this many of B walkers above the threshold: the batch goes to float64 whole (hysteresis: see refine_direct_enter)
  bool mostly_flagged(long n_above, int B) {
    const double lim = was_direct ? refine_direct_exit : refine_direct_enter;
    const bool yes = B >= 16 && (double)n_above > lim * (double)B;
    direct_streak = (yes && was_direct) ? direct_streak + 1 : 0;
    was_direct = yes;
    return yes;
  }
  // calls the context now stays in the direct mode: doubled by every look that confirmed it (refine_direct_backoff)
  int direct_stay() const {
    const int sh = direct_streak < refine_direct_backoff ? direct_streak : refine_direct_backoff;
    const long n = (long)refine_direct_calls << (sh < 16 ? sh : 16);
    return n > 1000000 ? 1000000 : (int)n;
  }
  int lap_refined_core_(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    last_refined = 0;
    ++refine_counters[0];
    if constexpr (sizeof(real) == 8) {
      return pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
    } else {
      if (!refine) return pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
      if ((size_t)B + 1 > flag_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_flag) { HIP_TRY(hipFree(d_flag)); d_flag = nullptr; }
        HIP_TRY(hipMalloc((void**)&d_flag, sizeof(int32_t) * ((size_t)B + 1)));
        flag_cap = (size_t)B + 1;
      }
      if ((size_t)B > score_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_score) { HIP_TRY(hipFree(d_score)); d_score = nullptr; }
        HIP_TRY(hipMalloc((void**)&d_score, sizeof(double) * (size_t)B));
        score_cap = (size_t)B;
      }
      int rc = DQMC_OK;
      // mode 1 on a system where most walkers get flagged (deep attention networks, ill-conditioned Slater matrices of a
      // random-init TransPsiformer): the float32 pass would be wasted, so the following 15 calls go to float64 directly,
      // then the float32 pass is probed again
      const bool direct = refine >= 2 || (refine == 1 && refine_all_calls > 0 && twin);
      if (refine == 1 && refine_all_calls > 0) --refine_all_calls;
      if (direct) {                // the whole forward-Laplacian pass in float64 (float32 stays the sampling dtype)
        ++refine_counters[1];
        last_score_B = 0;          // (no float32 pass, no scores)
        rc = ensure_twin();
        if (rc) return rc;
        std::vector<int32_t> iota((size_t)B);
        for (int k = 0; k < B; ++k) iota[k] = k;
        rc = upload_list(iota);
        if (rc) return rc;
        return refine_listed(r, R, B, B, e_loc, stats, grad, logpsi, sign);
      }
      HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int32_t), st));
      if (!d_thresh) HIP_TRY(hipMalloc((void**)&d_thresh, sizeof(double)));
      if (thresh_uploaded != refine_thresh) {      // (k_final reads the threshold from here: a captured pass must follow a re-calibration)
        HIP_TRY(hipMemcpyAsync(d_thresh, &refine_thresh, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        thresh_uploaded = refine_thresh;
      }
      flag_on = true;
      rc = pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
      flag_on = false;
      last_score_B = rc ? 0 : B;
      if (rc) return rc;
      const bool probe = refine == 1 && refine_probe > 0 && e_loc && (calls_since_probe < 0 || calls_since_probe + 1 >= refine_probe);
      if (calls_since_probe >= 0) ++calls_since_probe;
      int32_t n = 0;
      HIP_TRY(hipMemcpyAsync(&n, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (n > B) n = B;
      if (!probe) {
        if (n <= 0) { mostly_flagged(0, B); return DQMC_OK; }      // (a float32 look that flags nothing leaves the direct mode: reset the hysteresis state)
        rc = ensure_twin();
        if (rc == DQMC_E_UNSUPPORTED && refine == 1) { refine = 0; return DQMC_OK; }   // no float64 kernel set for this program: float32 stands
        if (rc) return rc;
        if (refine == 1 && mostly_flagged(n, B)) {
          // most of the batch is beyond float32: this call and the next 15 evaluate everything in float64
          refine_all_calls = direct_stay();
          ++refine_counters[1];
          std::vector<int32_t> iota((size_t)B);
          for (int k = 0; k < B; ++k) iota[k] = k;
          rc = upload_list(iota);
          if (rc) return rc;
          n = B;
        }
        // the twin's pass replays a captured graph per batch size: round the count up to a multiple of 64 (the surplus rows
        // re-evaluate the first flagged walker and are not written back: k_refine_gather / scatter read the count on the
        // device), so that a handful of sizes serve every step
        const bool padded = n < B && static_cast<Engine<double>*>(twin)->graph_fits((n + 63) / 64 * 64);
        int n_eval = padded ? (n + 63) / 64 * 64 : n;
        if (n_eval > B) n_eval = B;
        const int32_t* d_cnt = padded ? d_flag : nullptr;
        rc = refine_listed(r, R, B, n_eval, e_loc, stats, grad, logpsi, sign, d_cnt);
        last_refined = n;
        HIP_TRY(hipGetLastError());
        return rc;
      }
      // ---- probe call: measure the float32 error per unit of score on a strided sample, re-derive the threshold, and
      // apply it to THIS call as well (a caller that evaluates once gets the calibrated result)
      ++refine_counters[2];
      rc = ensure_twin();
      if (rc == DQMC_E_UNSUPPORTED && refine == 1) { refine = 0; return DQMC_OK; }
      if (rc) return rc;
      std::vector<int32_t> flagged((size_t)n);
      std::vector<double> score((size_t)B);
      std::vector<float> e32((size_t)B);
      if (n) HIP_TRY(hipMemcpyAsync(flagged.data(), d_flag + 1, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(score.data(), d_score, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(e32.data(), e_loc, sizeof(float) * (size_t)B, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      std::vector<char> done((size_t)B, 0);        // 1: evaluated in float64 by the first pass of this call (flagged or sample)
      for (int32_t b : flagged) if (b >= 0 && b < B) done[b] = 1;
      // calibration sample: a strided subset of the WHOLE batch (flagged or not: every walker is an (error, score) pair; a
      // system whose walkers all sit above the current threshold must still be able to move it)
      const int ns = B < refine_sample ? B : refine_sample;
      std::vector<int32_t> sample, list(flagged);
      std::vector<int> sample_pos;                 // position of each sample walker in `list`
      {
        std::vector<int> pos_of((size_t)B, -1);
        for (size_t k = 0; k < flagged.size(); ++k) if (flagged[k] >= 0 && flagged[k] < B) pos_of[flagged[k]] = (int)k;
        for (int j = 0; j < ns; ++j) {
          const int b = (int)((long)j * B / ns);
          if (pos_of[b] < 0) { pos_of[b] = (int)list.size(); list.push_back(b); done[b] = 1; }
          sample.push_back(b);
          sample_pos.push_back(pos_of[b]);
        }
      }
      // ONE float64 pass over flagged + sample; between its evaluation and its write-back the threshold is re-derived from
      // the sample, and only walkers above the NEW threshold are written back: what a walker's result is depends on its
      // score and the threshold alone -- never on having served as a calibration sample or on the threshold before the probe
      rc = upload_list(list);
      if (rc) return rc;
      probe_rethreshold = [&]() {
        std::vector<double> cs;
        for (size_t k = 0; k < sample.size(); ++k) {
          const int32_t b = sample[k];
          const double e64v = probe_sample_e[(size_t)sample_pos[k]];
          const double rel = std::fabs(e64v - (double)e32[b]) / std::fmax(1.0, std::fabs(e64v));
          if (std::isfinite(rel) && std::isfinite(score[b]) && score[b] > 0) cs.push_back(rel / score[b]);
        }
        if (cs.size() >= 2) {
          // scale m of the exponential error model err = m x score x xi (engine.hip, above refine_thresh): two robust
          // estimates from the sample's quantiles, the larger one
          std::sort(cs.begin(), cs.end());
          const double q50 = cs[(size_t)(0.5 * (cs.size() - 1) + 0.5)], q90 = cs[(size_t)(0.9 * (cs.size() - 1) + 0.5)];
          const double m = std::fmax(std::fmax(q50 / 0.6931471805599453, q90 / 2.302585092994046), 1e-14);
          probe_c = probe_c > 0 ? std::sqrt(probe_c * m) : m;         // geometric smoothing over the probes
          // the largest threshold whose kept walkers (scores of THIS batch, ascending) miss the tolerance at an expected
          // rate <= refine_miss:  (1 / B) sum_{s_i <= thr} exp(-tol / (m s_i)) <= refine_miss
          std::vector<double> ss;
          for (int b = 0; b < B; ++b) if (std::isfinite(score[b]) && score[b] > 0) ss.push_back(score[b]);
          std::sort(ss.begin(), ss.end());
          double acc = 0.0, thr = 1.0;
          for (size_t k = 0; k < ss.size(); ++k) {
            acc += std::exp(-refine_target / (probe_c * ss[k])) / (double)B;
            if (acc > refine_miss) break;
            thr = ss[k];
          }
          if (!ss.empty() && acc <= refine_miss) thr = 1e9;           // (the whole batch may stay in float32)
          refine_thresh = std::fmin(std::fmax(thr, 1.0), 1e9);
        }
      };
      rc = refine_listed(r, R, B, (int)list.size(), e_loc, stats, grad, logpsi, sign, nullptr, nullptr, 0, true);
      probe_rethreshold = nullptr;
      if (rc) return rc;
      calls_since_probe = 0;
      long n_above = 0;
      for (int b = 0; b < B; ++b) if (!(score[b] <= refine_thresh)) ++n_above;
      std::vector<int32_t> more;
      if (refine == 1 && mostly_flagged(n_above, B)) {
        // most of the batch is beyond float32: the next calls go to float64 directly, and so does the rest of this one
        // (the few walkers below the threshold of such a system are not reliably predicted either)
        refine_all_calls = direct_stay();
        ++refine_counters[1];
        for (int b = 0; b < B; ++b) if (!done[b] || score[b] <= refine_thresh) more.push_back(b);     // not yet written back
        last_refined = B - (int)more.size();
      } else {
        for (int b = 0; b < B; ++b) {
          if (score[b] <= refine_thresh) continue;
          if (done[b]) ++last_refined; else more.push_back(b);
        }
      }
      if (!more.empty()) {
        rc = upload_list(more);
        if (rc) return rc;
        rc = refine_listed(r, R, B, (int)more.size(), e_loc, stats, grad, logpsi, sign);
        if (rc) return rc;
      }
      HIP_TRY(hipGetLastError());
      return DQMC_OK;
    }
  }

  // Effective core potentials: host tables (ecp/gaussian_type_ecp.py:32-93 layout) -> device.
  //   loc[n_nuc][3][2][n_t_loc]  r^-1 / r^0 / r^1 terms: [.,term,0,.] exponents, [.,term,1,.] coefficients
  //   nl [n_nuc][n_l][2][n_t_nl] channels l = 0..n_l-1; nuclei whose block is all zero have no non-local part
