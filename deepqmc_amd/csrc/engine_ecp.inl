// engine_ecp.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// effective core potentials and pseudo-Hamiltonian: tables -> device, the 12-point quadrature through the value path,
// the mixed-precision quadrature of float32 contexts.

  int ecp_mixed(const float* r, const float* R, int B, float* e_loc, float* stats) {
    // No float64 twin for this program (its float64 kernel set does not exist: DQMC_E_UNSUPPORTED -- lap_refined_core has
    // switched the refinement off for the same reason): the quadrature runs entirely in float32, as local_energy_ecp would
    // (every kept pair in the float32 class; the pair cut-off stays).  Any other failure is an error of the call.
    int rc = ensure_twin();
    const bool have_twin = rc == DQMC_OK;
    if (rc && rc != DQMC_E_UNSUPPORTED) return rc;
    const size_t per_walker = (size_t)ecp_n_nl * N * 12, triples_pw = (size_t)ecp_n_nl * N;
    int nbw = (int)(ecp_max_cfg / per_walker);
    nbw = nbw < 1 ? 1 : (nbw > B ? B : nbw);
    const size_t n_cfg = (size_t)nbw * (per_walker + 1), n_tr = (size_t)nbw * triples_pw;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_cls = 0, o_ll = o_cls + al(4 * n_tr), o_lh = o_ll + al(4 * n_tr), o_cnt = o_lh + al(4 * n_tr),
                 o_r32 = o_cnt + 256, o_l32 = o_r32 + al(4 * n_cfg * N * 3), o_s32 = o_l32 + al(4 * n_cfg),
                 o_r64 = o_s32 + al(4 * n_cfg), o_l64 = o_r64 + al(8 * n_cfg * N * 3), o_s64 = o_l64 + al(8 * n_cfg),
                 o_R64 = o_s64 + al(4 * n_cfg), tot = o_R64 + al(8 * 3 * (size_t)sys.n_nuc);
    if (tot > ecpm_bytes) {
      if (d_ecpm) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ecpm)); d_ecpm = nullptr; ecpm_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ecpm, tot);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "ECP scratch of " + std::to_string(tot) + " bytes: " + hipGetErrorString(e));
      ecpm_bytes = tot;
    }
    int32_t* cls = (int32_t*)(d_ecpm + o_cls); int32_t* list_l = (int32_t*)(d_ecpm + o_ll); int32_t* list_h = (int32_t*)(d_ecpm + o_lh);
    int32_t* cnt = (int32_t*)(d_ecpm + o_cnt);
    float* rq32 = (float*)(d_ecpm + o_r32); float* lq32 = (float*)(d_ecpm + o_l32); int32_t* sq32 = (int32_t*)(d_ecpm + o_s32);
    double* rq64 = (double*)(d_ecpm + o_r64); double* lq64 = (double*)(d_ecpm + o_l64); int32_t* sq64 = (int32_t*)(d_ecpm + o_s64);
    double* R64 = (double*)(d_ecpm + o_R64);
    if (have_twin) dqmc::launch_refine_gather(st, r, R, nullptr, nullptr, 0, 3 * N, 3 * sys.n_nuc, nullptr, R64);      // (widens R only)
    dqmc::EcpMixArgs a{};
    a.r = r; a.R = R; a.nl_nuc = d_ecp_nuc; a.nl = d_ecp_nl; a.phi = (const float*)ecp_phi; a.seed = ecp_seed;
    a.B = B; a.N = N; a.n_nl = ecp_n_nl; a.L = ecp_L; a.n_t = ecp_nt_nl;
    a.w_heavy = have_twin ? ecp_w_heavy : HUGE_VAL; a.w_skip = ecp_w_skip;
    a.dlog_floor = ecp_dlog_floor;
    ecp_last_counts[0] = ecp_last_counts[1] = ecp_last_counts[2] = 0;
    for (int b0 = 0; b0 < B; b0 += nbw) {
      a.b0 = b0; a.nb = (B - b0) < nbw ? (B - b0) : nbw;
      HIP_TRY(hipMemsetAsync(cnt, 0, 2 * sizeof(int32_t), st));
      a.l32 = nullptr;
      if (have_twin && ecp_dlog_floor > 0) {
        // psi(r) of the chunk's own walkers by both value paths, ahead of the classification (the lists are empty: the
        // configurations are the nb walkers themselves); their disagreement is each walker's float32 error
        t_begin("ecp", 0);
        dqmc::launch_ecp_points_list<float>(st, a, list_l, 0, rq32);
        dqmc::launch_ecp_points_list<double>(st, a, list_h, 0, rq64);
        t_end();
        rc = run((const real*)rq32, (const real*)R, a.nb, false, (real*)lq32, sq32, nullptr, nullptr, nullptr);
        if (rc) return rc;
        rc = twin->wf_eval(rq64, R64, a.nb, lq64, sq64);
        if (rc) return rc;
        a.l32 = lq32; a.l64 = lq64; a.s32 = sq32; a.s64 = sq64;
      }
      t_begin("ecp", 0);
      dqmc::launch_ecp_classify(st, a, cls, list_l, list_h, cnt);
      t_end();
      int32_t n2[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(n2, cnt, sizeof(n2), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      ecp_last_counts[0] += n2[0]; ecp_last_counts[1] += n2[1];
      ecp_last_counts[2] += (long)a.nb * (long)triples_pw - n2[0] - n2[1];
      t_begin("ecp", 0);
      dqmc::launch_ecp_points_list<float>(st, a, list_l, n2[0], rq32);
      if (have_twin) dqmc::launch_ecp_points_list<double>(st, a, list_h, n2[1], rq64);
      t_end();
      rc = run((const real*)rq32, (const real*)R, a.nb + 12 * n2[0], false, (real*)lq32, sq32, nullptr, nullptr, nullptr);
      if (rc) return rc;
      if (have_twin) {
        rc = twin->wf_eval(rq64, R64, a.nb + 12 * n2[1], lq64, sq64);
        if (rc) return rc;
      } else if (n2[1] != 0) {
        return fail(DQMC_E_HIP, "ECP classification produced float64 pairs without a float64 twin");
      }
      t_begin("ecp", 0);
      dqmc::launch_ecp_reduce_mixed(st, a, cls, lq32, sq32, lq64, sq64, e_loc, stats);
      t_end();
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int set_ecp(int n_t_loc, const double* loc, int n_l, int n_t_nl, const double* nl) override {
    if (n_t_loc < 0 || n_l < 0 || n_t_nl < 0) return fail(DQMC_E_ARG, "negative ECP table size");
    ++graph_epoch; drop_graphs();          // (captured passes hold the table pointers)
    if (ph_n && nl && n_l > 0 && n_t_nl > 0) return fail(DQMC_E_ARG, "a pseudo-Hamiltonian and a non-local Gaussian ECP cannot both be set");
    HIP_TRY(hipStreamSynchronize(st));
    if (d_ecp_loc) { HIP_TRY(hipFree(d_ecp_loc)); d_ecp_loc = nullptr; }
    if (d_ecp_nl) { HIP_TRY(hipFree(d_ecp_nl)); d_ecp_nl = nullptr; }
    if (d_ecp_nuc) { HIP_TRY(hipFree(d_ecp_nuc)); d_ecp_nuc = nullptr; }
    ecp_nt_loc = ecp_n_nl = ecp_L = ecp_nt_nl = 0;
    ecp_loc_h.clear(); ecp_loc_nt_h = 0;
    ecp_nl_h.clear(); ecp_nl_L_h = ecp_nl_nt_h = 0;
    if (loc && n_t_loc > 0 && sizeof(real) == 4) { ecp_loc_h.assign(loc, loc + (size_t)sys.n_nuc * 6 * n_t_loc); ecp_loc_nt_h = n_t_loc; }
    if (nl && n_l > 0 && n_t_nl > 0 && sizeof(real) == 4) { ecp_nl_h.assign(nl, nl + (size_t)sys.n_nuc * n_l * 2 * n_t_nl); ecp_nl_L_h = n_l; ecp_nl_nt_h = n_t_nl; }
    if (twin) { const int rc = twin->set_ecp(n_t_loc, loc, n_l, n_t_nl, nl); if (rc) return rc; }
    if (loc && n_t_loc > 0) {
      const size_t n = (size_t)sys.n_nuc * 6 * n_t_loc;
      HIP_TRY(hipMalloc((void**)&d_ecp_loc, sizeof(double) * n));
      HIP_TRY(hipMemcpy(d_ecp_loc, loc, sizeof(double) * n, hipMemcpyHostToDevice));
      ecp_nt_loc = n_t_loc;
    }
    if (nl && n_l > 0 && n_t_nl > 0) {
      const size_t blk = (size_t)n_l * 2 * n_t_nl;
      std::vector<int32_t> nuc;
      std::vector<double> compact;
      for (int a = 0; a < sys.n_nuc; ++a) {          // gaussian_type_ecp.py:121 (nuc_with_nl_pot)
        bool any = false;
        for (size_t k = 0; k < blk; ++k) any = any || nl[a * blk + k] != 0.0;
        if (!any) continue;
        nuc.push_back(a);
        compact.insert(compact.end(), nl + a * blk, nl + (a + 1) * blk);
      }
      if (!nuc.empty()) {
        HIP_TRY(hipMalloc((void**)&d_ecp_nl, sizeof(double) * compact.size()));
        HIP_TRY(hipMalloc((void**)&d_ecp_nuc, sizeof(int32_t) * nuc.size()));
        HIP_TRY(hipMemcpy(d_ecp_nl, compact.data(), sizeof(double) * compact.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_ecp_nuc, nuc.data(), sizeof(int32_t) * nuc.size(), hipMemcpyHostToDevice));
        ecp_n_nl = (int)nuc.size(); ecp_L = n_l; ecp_nt_nl = n_t_nl;
      }
    }
    return DQMC_OK;
  }
  int ecp_rotation(uint64_t seed, const void* phi) override { ecp_seed = seed; ecp_phi = phi; return DQMC_OK; }

  // Pseudo-Hamiltonian tables (ecp/pseudo_hamiltonian.py:73-112): rV_loc and rV_L2 on the regular grid
  // linspace(0, r_max, n_grid), one row per nucleus; rows of nuclei with mask 0 are ignored.  An all-zero mask
  // (or n_grid 0) switches the PH off.
  int set_ph(int n_grid, double r_max, const double* rv_loc, const double* rv_l2, const int32_t* mask) override {
    ++graph_epoch; drop_graphs();
    HIP_TRY(hipStreamSynchronize(st));
    if (d_ph_loc) { HIP_TRY(hipFree(d_ph_loc)); d_ph_loc = nullptr; }
    if (d_ph_l2) { HIP_TRY(hipFree(d_ph_l2)); d_ph_l2 = nullptr; }
    if (d_ph_nuc) { HIP_TRY(hipFree(d_ph_nuc)); d_ph_nuc = nullptr; }
    ph_n = ph_grid = 0; ph_rmax = 0.0;
    ph_loc_h.clear(); ph_l2_h.clear(); ph_mask_h.clear();
    std::vector<int32_t> nuc;
    if (mask && n_grid > 0) for (int a = 0; a < sys.n_nuc; ++a) if (mask[a]) nuc.push_back(a);
    if (nuc.empty()) { if (twin) return twin->set_ph(0, 0.0, nullptr, nullptr, nullptr); return DQMC_OK; }
    if (n_grid < 2 || !(r_max > 0.0) || !rv_loc || !rv_l2) return fail(DQMC_E_ARG, "pseudo-Hamiltonian tables need n_grid >= 2, r_max > 0");
    if (ecp_n_nl) return fail(DQMC_E_ARG, "a pseudo-Hamiltonian and a non-local Gaussian ECP cannot both be set");
    std::vector<double> loc, l2;
    for (int a : nuc) {
      loc.insert(loc.end(), rv_loc + (size_t)a * n_grid, rv_loc + (size_t)(a + 1) * n_grid);
      l2.insert(l2.end(), rv_l2 + (size_t)a * n_grid, rv_l2 + (size_t)(a + 1) * n_grid);
    }
    HIP_TRY(hipMalloc((void**)&d_ph_loc, sizeof(double) * loc.size()));
    HIP_TRY(hipMalloc((void**)&d_ph_l2, sizeof(double) * l2.size()));
    HIP_TRY(hipMalloc((void**)&d_ph_nuc, sizeof(int32_t) * nuc.size()));
    HIP_TRY(hipMemcpy(d_ph_loc, loc.data(), sizeof(double) * loc.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ph_l2, l2.data(), sizeof(double) * l2.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ph_nuc, nuc.data(), sizeof(int32_t) * nuc.size(), hipMemcpyHostToDevice));
    ph_n = (int)nuc.size(); ph_grid = n_grid; ph_rmax = r_max;
    if (sizeof(real) == 4) {
      ph_loc_h.assign(rv_loc, rv_loc + (size_t)sys.n_nuc * n_grid);
      ph_l2_h.assign(rv_l2, rv_l2 + (size_t)sys.n_nuc * n_grid);
      ph_mask_h.assign(mask, mask + sys.n_nuc);
    }
    if (twin) return twin->set_ph(n_grid, r_max, rv_loc, rv_l2, mask);
    return DQMC_OK;
  }

  // E_loc with the non-local ECP term: the Laplacian pass, then 12 N n_nl value-only psi evaluations per
  // walker in batches of <= ecp_max_cfg quadrature walkers (gaussian_type_ecp.py:161-255).
  int local_energy_ecp(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi,
                       int32_t* sign) {
    const size_t per_walker = (size_t)ecp_n_nl * N * 12;
    int nbw = (int)(ecp_max_cfg / per_walker);
    nbw = nbw < 1 ? 1 : (nbw > B ? B : nbw);
    const size_t n_cfg = (size_t)nbw * per_walker;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rq = 0, o_lq = o_rq + al(sizeof(real) * n_cfg * N * 3), o_sq = o_lq + al(sizeof(real) * n_cfg),
                 o_l0 = o_sq + al(sizeof(int32_t) * n_cfg), o_s0 = o_l0 + al(sizeof(real) * B),
                 tot = o_s0 + al(sizeof(int32_t) * B);
    if (tot > ecp_bytes) {
      if (d_ecp) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ecp)); d_ecp = nullptr; ecp_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ecp, tot);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "ECP scratch of " + std::to_string(tot) + " bytes: " + hipGetErrorString(e));
      ecp_bytes = tot;
    }
    real* rq = (real*)(d_ecp + o_rq); real* lq = (real*)(d_ecp + o_lq); int32_t* sq = (int32_t*)(d_ecp + o_sq);
    real* l0 = logpsi ? logpsi : (real*)(d_ecp + o_l0);
    int32_t* s0 = sign ? sign : (int32_t*)(d_ecp + o_s0);
    int rc = run(r, R, B, true, l0, s0, e_loc, stats, grad);
    if (rc) return rc;
    dqmc::EcpArgs a{};
    a.r = r; a.R = R; a.nl_nuc = d_ecp_nuc; a.nl = d_ecp_nl; a.phi = ecp_phi; a.seed = ecp_seed;
    a.walker_idx = ecp_idx; a.phi_f32 = ecp_phi_f32 ? 1 : 0;
    a.B = B; a.N = N; a.n_nl = ecp_n_nl; a.L = ecp_L; a.n_t = ecp_nt_nl;
    for (int b0 = 0; b0 < B; b0 += nbw) {
      a.b0 = b0; a.nb = (B - b0) < nbw ? (B - b0) : nbw;
      t_begin("ecp", 0);
      dqmc::launch_ecp_points<real>(st, a, rq);
      t_end();
      rc = run(rq, R, (int)((size_t)a.nb * per_walker), false, lq, sq, nullptr, nullptr, nullptr);
      if (rc) return rc;
      t_begin("ecp", 0);
      dqmc::launch_ecp_reduce<real>(st, a, lq, sq, l0, s0, e_loc, stats, (real*)nullptr);
      t_end();
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
