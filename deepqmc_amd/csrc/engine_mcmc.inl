// engine_mcmc.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// Metropolis / Langevin / exchange drivers, energy records, dqmc_debug_read.

  // algorithmic flops of the linear layers of one value-only evaluation, per walker (the count run_fused2 reports)
  double substep_flops() const {
    double flops = 0;
    for (int k = 0; k < fused_n_ops; ++k)
      if (ops[k].kind == DQMC_OP_LINEAR) {
        int ktot = 0;
        for (int p = 0; p < ops[k].i[0]; ++p) ktot += ops[k].i[3 + 4 * p];
        flops += 2.0 * ops[k].i[20] * (double)ktot * ops[k].i[21];
      }
    return flops;
  }
  int substep_kernel(char* name_out, size_t n) override {
    name_out[0] = 0;
    if (!(spec_k && fused_spec && d_tape)) return 0;
    snprintf(name_out, n, "k_substep_%s", spec_k->name);
    return 1;
  }
  int mcmc(void* r_, void* logpsi_, int32_t* sign, int32_t* age, void* tau_, const void* R_, int B, int n_sub,
           int max_age, double target, uint64_t seed, const void* noise_, const void* unif_, uint8_t* accept_out,
           double* stats7) override {
    if (B < 1 || n_sub < 0) return fail(DQMC_E_ARG, "bad B / n_sub");
    if ((noise_ == nullptr) != (unif_ == nullptr)) return fail(DQMC_E_ARG, "noise and unif must both be given or both NULL");
    real* r = (real*)r_; real* logpsi = (real*)logpsi_; real* tau = (real*)tau_;
    const real* R = (const real*)R_;
    const size_t n_r = (size_t)B * N * 3;
    // scratch: r_prop, logpsi_prop, sign_prop, noise, unif
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_lp = o_rp + al(sizeof(real) * n_r), o_sp = o_lp + al(sizeof(real) * B),
                 o_nz = o_sp + al(sizeof(int32_t) * B), o_un = o_nz + al(sizeof(real) * n_r * (size_t)(n_sub > 0 ? n_sub : 1)),
                 tot = o_un + al(sizeof(real) * B * (size_t)(n_sub > 0 ? n_sub : 1));
    if (tot > mc_bytes) {
      if (d_mc) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_mc)); d_mc = nullptr; }
      HIP_TRY(hipMalloc((void**)&d_mc, tot));
      mc_bytes = tot;
    }
    real* r_prop = (real*)(d_mc + o_rp); real* lp_prop = (real*)(d_mc + o_lp);
    int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    real* nz = (real*)(d_mc + o_nz); real* un = (real*)(d_mc + o_un);
    if (!noise_ && n_sub > 0) {      // all sub-steps' normals and uniforms in ONE launch (Philox is counter based)
      t_begin("mcmc", 0);
      dqmc::launch_rng<real>(st, nz, (long)(n_r * n_sub), un, (long)B * n_sub, seed, (uint64_t)0);
      t_end();
    }
    // whole sub-step in one launch (kernel_fused2.hip: propose in the prologue, determinants / CI sum / accept /
    // tau adaptation in the tail) when the ansatz tail is the plain SLOGDET + FINAL pair and N <= 4
    const bool one_launch = fused_enabled && fused2_WT > 0 && fused_substep && N >= 2 && N <= 4 &&
                            sys.n_nuc <= 8 && fused2_WT <= 16 && (int)ops.size() == fused_n_ops + 2 &&
                            ops[fused_n_ops].kind == DQMC_OP_SLOGDET && ops[fused_n_ops + 1].kind == DQMC_OP_FINAL &&
                            substep_mat_off() >= 0;
    for (int s = 0; s < n_sub; ++s) {
      const real* noise_s; const real* unif_s;
      if (one_launch) {
        if (noise_) {
          noise_s = (const real*)noise_ + (size_t)s * n_r;
          unif_s = (const real*)unif_ + (size_t)s * B;
        } else {
          noise_s = nz + (size_t)s * n_r; unif_s = un + (size_t)s * B;
        }
        dqmc::LaneInfo li; li.N = N; li.T = 1; li.TP = 1;
        int rc = plan(B, 1);
        if (rc) return rc;
        const dqmc_op& fin = ops[fused_n_ops + 1];
        dqmc::FusedMc mc{};
        mc.enabled = 1; mc.noise = noise_s; mc.unif = unif_s; mc.r = r; mc.logpsi = logpsi; mc.sign = sign; mc.age = age;
        mc.tau_in = tau; mc.tau_ring = d_tau_ring; mc.counters = d_nacc; mc.s = s; mc.target = target;
        mc.accept_out = accept_out ? accept_out + (size_t)s * B : nullptr;
        mc.max_age = max_age;
        mc.orb_op = -1;
        for (int j = 0; j < fused_n_ops; ++j) if (ops[f_order[j]].kind == DQMC_OP_ORBITALS) mc.orb_op = j;
        mc.mat_off = substep_mat_off();
        mc.jas_width = fin.i[0] >= 0 ? bufs[fin.i[0]].width : 0;
        mc.cc_off = fin.i[1]; mc.cusp_kind = fin.i[2]; mc.al_off = fin.i[3];
        mc.same_scale = fin.f[0]; mc.anti_scale = fin.f[1];
        if (spec_k && fused_spec && d_tape) {
          // the plan-specialised kernel of this program (csrc/gen, spec_device.h): one wave per tile of 16 / N walkers
          if constexpr (sizeof(real) == 4) {
            dqmc::SpecArgs sa{};
            sa.tape = d_tape; sa.w = reinterpret_cast<const float*>(d_w); sa.R = reinterpret_cast<const float*>(R);
            sa.B = B; sa.eps = sys.norm_eps; sa.mc = mc;
            sa.prof = fused_dbg ? d_prof : nullptr;
            t_begin("fused_substep", (double)B * substep_flops());
            spec_k->launch(st, sa, (B + spec_k->walkers_per_block - 1) / spec_k->walkers_per_block);
            t_end();
          }
        } else {
          rc = run_fused2(nullptr, R, B, li, &mc);
          if (rc) return rc;
        }
        if (s + 1 == n_sub) {
          t_begin("mcmc", 0);
          dqmc::launch_tau_finalize<real>(st, tau, (const real*)d_tau_ring, d_nacc, s, B, target, d_acc);
          t_end();
        }
        continue;
      }
      if (noise_) {
        noise_s = (const real*)noise_ + (size_t)s * n_r;
        unif_s = (const real*)unif_ + (size_t)s * B;
      } else {
        noise_s = nz + (size_t)s * n_r; unif_s = un + (size_t)s * B;
      }
      t_begin("mcmc", 0);
      dqmc::launch_propose<real>(st, r, noise_s, tau, r_prop, (long)n_r);
      t_end();
      int rc = run(r_prop, R, B, false, lp_prop, s_prop, nullptr, nullptr, nullptr);
      if (rc) return rc;
      t_begin("mcmc", 0);
      dqmc::launch_accept<real>(st, r, logpsi, sign, age, r_prop, lp_prop, s_prop, unif_s, max_age, B, N, d_nacc,
                                accept_out ? accept_out + (size_t)s * B : nullptr);
      dqmc::launch_tau_update<real>(st, tau, d_nacc, B, target, d_acc);
      t_end();
    }
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, tau, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }

  // ---- Langevin (MALA) and exchange steps (electron_samplers.py:176-330, sampling_utils.py:72-101) ----
  double* d_molz = nullptr;     // device copy of the full nuclear charges the drift cleaning uses
  int upload_molz(const double* z) {
    if (!d_molz) HIP_TRY(hipMalloc((void**)&d_molz, sizeof(double) * sys.n_nuc));
    HIP_TRY(hipMemcpyAsync(d_molz, z, sizeof(double) * sys.n_nuc, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));     // `z` is pageable host memory of the caller
    return DQMC_OK;
  }
  int ensure_mc(size_t tot) {
    if (tot > mc_bytes) {
      if (d_mc) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_mc)); d_mc = nullptr; }
      HIP_TRY(hipMalloc((void**)&d_mc, tot));
      mc_bytes = tot;
    }
    return DQMC_OK;
  }
  int langevin_update(const void* r_, const void* R_, const double* molz, int B, const void* tau_, void* logpsi, int32_t* sign,
                      void* force_) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    int rc = upload_molz(molz);
    if (rc) return rc;
    const size_t n_r = (size_t)B * N * 3;
    rc = ensure_mc(sizeof(real) * n_r + 256);
    if (rc) return rc;
    real* g = (real*)d_mc;
    rc = lap_refined((const real*)r_, (const real*)R_, B, nullptr, nullptr, g, (real*)logpsi, sign);
    if (rc) return rc;
    t_begin("mcmc", 0);
    dqmc::launch_clean_force<real>(st, g, (const real*)r_, (const real*)R_, d_molz, (const real*)tau_, B, N, sys.n_nuc, (real*)force_);
    t_end();
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int langevin(void* r_, void* logpsi_, int32_t* sign, int32_t* age, void* force_, void* tau_, const void* R_, const double* molz,
               int B, int n_sub, int max_age, double target, uint64_t seed, const void* noise_, const void* unif_,
               uint8_t* accept_out, double* stats7) override {
    if (B < 1 || n_sub < 0) return fail(DQMC_E_ARG, "bad B / n_sub");
    if ((noise_ == nullptr) != (unif_ == nullptr)) return fail(DQMC_E_ARG, "noise and unif must both be given or both NULL");
    int rc = upload_molz(molz);
    if (rc) return rc;
    real* r = (real*)r_; real* logpsi = (real*)logpsi_; real* tau = (real*)tau_; real* force = (real*)force_;
    const real* R = (const real*)R_;
    const size_t n_r = (size_t)B * N * 3, ns = (size_t)(n_sub > 0 ? n_sub : 1);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_g = o_rp + al(sizeof(real) * n_r), o_fp = o_g + al(sizeof(real) * n_r), o_lp = o_fp + al(sizeof(real) * n_r),
                 o_sp = o_lp + al(sizeof(real) * B), o_nz = o_sp + al(sizeof(int32_t) * B), o_un = o_nz + al(sizeof(real) * n_r * ns),
                 tot = o_un + al(sizeof(real) * B * ns);
    rc = ensure_mc(tot);
    if (rc) return rc;
    real* r_prop = (real*)(d_mc + o_rp); real* g_prop = (real*)(d_mc + o_g); real* f_prop = (real*)(d_mc + o_fp);
    real* lp_prop = (real*)(d_mc + o_lp); int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    real* nz = (real*)(d_mc + o_nz); real* un = (real*)(d_mc + o_un);
    if (!noise_ && n_sub > 0) {
      t_begin("mcmc", 0);
      dqmc::launch_rng<real>(st, nz, (long)(n_r * n_sub), un, (long)B * n_sub, seed, (uint64_t)1);
      t_end();
    }
    for (int s = 0; s < n_sub; ++s) {
      const real* noise_s = noise_ ? (const real*)noise_ + (size_t)s * n_r : nz + (size_t)s * n_r;
      const real* unif_s = unif_ ? (const real*)unif_ + (size_t)s * B : un + (size_t)s * B;
      t_begin("mcmc", 0);
      dqmc::launch_langevin_propose<real>(st, r, force, noise_s, tau, r_prop, (long)n_r);
      t_end();
      rc = lap_refined(r_prop, R, B, nullptr, nullptr, g_prop, lp_prop, s_prop);
      if (rc) return rc;
      t_begin("mcmc", 0);
      dqmc::launch_clean_force<real>(st, g_prop, r_prop, R, d_molz, tau, B, N, sys.n_nuc, f_prop);
      dqmc::launch_langevin_accept<real>(st, r, logpsi, sign, age, force, r_prop, lp_prop, s_prop, f_prop, unif_s, tau, max_age, B, N,
                                         d_nacc, accept_out ? accept_out + (size_t)s * B : nullptr);
      dqmc::launch_tau_update<real>(st, tau, d_nacc, B, target, d_acc);
      t_end();
    }
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, tau, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int exchange(void* r_, void* logpsi_, int32_t* sign, int32_t* age, const void* tau_, const void* R_, int B, const int32_t* up_idx,
               const int32_t* down_idx, const void* unif_, uint8_t* accept_out, double* stats7) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    if (sys.n_up < 1 || sys.n_down < 1) return fail(DQMC_E_ARG, "an exchange step needs electrons of both spins");
    real* r = (real*)r_; real* logpsi = (real*)logpsi_;
    const size_t n_r = (size_t)B * N * 3;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_lp = o_rp + al(sizeof(real) * n_r), o_sp = o_lp + al(sizeof(real) * B), tot = o_sp + al(sizeof(int32_t) * B);
    int rc = ensure_mc(tot);
    if (rc) return rc;
    real* r_prop = (real*)(d_mc + o_rp); real* lp_prop = (real*)(d_mc + o_lp); int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    t_begin("mcmc", 0);
    dqmc::launch_exchange_propose<real>(st, r, up_idx, down_idx, sys.n_up, B, N, r_prop);
    t_end();
    rc = run(r_prop, (const real*)R_, B, false, lp_prop, s_prop, nullptr, nullptr, nullptr);
    if (rc) return rc;
    t_begin("mcmc", 0);
    // `_accept` without max_age / target_acceptance (electron_samplers.py:312-313): no age override, tau unchanged
    dqmc::launch_accept<real>(st, r, logpsi, sign, age, r_prop, lp_prop, s_prop, (const real*)unif_, -1, B, N, d_nacc, accept_out);
    dqmc::launch_read_accept(st, d_nacc, B, d_acc);
    t_end();
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, (const real*)tau_, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }

  int energy_stats(const void* e, const void* w, int B, double* out7) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    dqmc::launch_energy_stats<real>(st, (const real*)e, (const real*)w, B, d_acc + 8);
    HIP_TRY(hipMemcpyAsync(out7, d_acc + 8, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }

  int energy_stats_dev(const void* e, const void* w, int B, double** rec_dev) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    dqmc::launch_energy_stats<real>(st, (const real*)e, (const real*)w, B, d_acc + 8);
    *rec_dev = d_acc + 8;
    return DQMC_OK;
  }

  int debug_read(int buf, double* out, size_t n) override {
    if (last_B == 0) return fail(DQMC_E_ARG, "no evaluation has run yet");
    HIP_TRY(hipStreamSynchronize(st));
    if constexpr (sizeof(real) == 4) {
      // the last Laplacian-mode evaluation handed its tail to the float64 twin: what the tail wrote lives in the twin's workspace
      if (last_tail && last_TP > 1 && twin && (buf == -1 || buf == -2 || buf == -4 || (buf >= 0 && buf < (int)bufs.size() && tail_written[buf]))) {
        auto* tw = static_cast<Engine<double>*>(twin);
        if (tw->last_B == last_B && tw->last_TP == last_TP) return tw->debug_read(buf, out, n);
        return fail(DQMC_E_ARG, "the float64 twin has been re-planned since the tail of this evaluation ran (a refinement pass): read with \"refine\" 0");
      }
    }
    if (buf == -1) {
      const size_t cnt = (size_t)last_B * sys.n_det * last_TP;
      if (n != cnt) return fail(DQMC_E_ARG, "size mismatch");
      HIP_TRY(hipMemcpy(out, d_ws + off_logdet, sizeof(double) * cnt, hipMemcpyDeviceToHost));
      return DQMC_OK;
    }
    if (buf == -4) {   // conditioning record per walker of the last Laplacian-mode evaluation
      if (n != (size_t)last_B || last_TP == 1) return fail(DQMC_E_ARG, "size mismatch or no Laplacian-mode evaluation");
      HIP_TRY(hipMemcpy(out, d_ws + off_kappa, sizeof(double) * n, hipMemcpyDeviceToHost));
      return DQMC_OK;
    }
    if (buf == -3) {   // per-op shader-clock stamps of the fused kernel (workgroup 0)
      if (!d_prof || n > 9 * ops.size() + 80 + 1024 + 2 * 8192) return fail(DQMC_E_ARG, "profile not enabled or size mismatch");
      std::vector<long long> tmp(n);
      HIP_TRY(hipMemcpy(tmp.data(), d_prof, sizeof(long long) * n, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; ++k) out[k] = (double)tmp[k];
      return DQMC_OK;
    }
    if (buf == -2) {
      const size_t cnt = (size_t)last_B * sys.n_det;
      if (n != cnt) return fail(DQMC_E_ARG, "size mismatch");
      std::vector<int32_t> tmp(cnt);
      HIP_TRY(hipMemcpy(tmp.data(), d_ws + off_signk, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < cnt; ++k) out[k] = tmp[k];
      return DQMC_OK;
    }
    if (buf < 0 || buf >= (int)bufs.size()) return fail(DQMC_E_ARG, "no such buffer");
    const size_t cnt = (size_t)last_B * bufs[buf].rows * last_TP * bufs[buf].width;
    if (n != cnt) return fail(DQMC_E_ARG, "size mismatch: expected " + std::to_string(cnt));
    const int lanes = lanes_of(buf, last_TP);
    const size_t cnt_dev = (size_t)last_B * bufs[buf].rows * lanes * bufs[buf].width;
    std::vector<real> tmp(cnt_dev);
    HIP_TRY(hipMemcpy(tmp.data(), d_ws + buf_off[buf], sizeof(real) * cnt_dev, hipMemcpyDeviceToHost));
    if (lanes == last_TP) {
      for (size_t k = 0; k < cnt; ++k) out[k] = (double)tmp[k];
      return DQMC_OK;
    }
    // pair-compact buffer: expand to the documented full-lane layout [B][rows][TP][width]
    const int T = 3 * N + 2, W = bufs[buf].width, rows = bufs[buf].rows;
    std::fill(out, out + cnt, 0.0);
    for (int b = 0; b < last_B; ++b)
      for (int row = 0; row < rows; ++row) {
        const int rc = pair_rs[buf][2 * row], sd = pair_rs[buf][2 * row + 1];
        for (int ct = 0; ct < lanes; ++ct) {
          int t;
          if (ct == 0) t = 0;
          else if (ct == lanes - 1) t = T - 1;
          else if (ct < 4) t = 1 + 3 * rc + (ct - 1);
          else if (sd < 0 || sd == rc) continue;
          else t = 1 + 3 * sd + (ct - 4);
          const real* src = tmp.data() + (((size_t)b * rows + row) * lanes + ct) * W;
          double* dst = out + (((size_t)b * rows + row) * last_TP + t) * W;
          for (int c = 0; c < W; ++c) dst[c] = (double)src[c];
        }
      }
    return DQMC_OK;
  }
