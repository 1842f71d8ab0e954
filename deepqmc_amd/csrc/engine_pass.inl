// engine_pass.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// the pass runner: workspace planning, walker chunks, the forward-Laplacian / value pass op by op over up to four HIP
// streams, hipGraph capture and replay; wf_eval / local_energy / psi_grad entry points.

  static bool lanes_supported(int TP) {
    return TP == 1 || TP == 16 || TP == 32 || TP == 48 || TP == 64 || TP == 96 || TP == 128;
  }

  // `only` (a twin planning for a float64 tail): memory for these buffers alone; the others keep offset 0 and are not touched
  int plan(int B, int TP, const std::vector<char>* only = nullptr) {
    buf_off.resize(bufs.size());
    size_t off = 0;
    auto bump = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    for (size_t k = 0; k < bufs.size(); ++k)
      buf_off[k] = (only && !(*only)[k]) ? 0 : bump(sizeof(real) * (size_t)B * bufs[k].rows * lanes_of((int)k, TP) * bufs[k].width);
    {
      int max_ldw = 4;
      for (const auto& o : ops) if (o.kind == DQMC_OP_LINEAR && pad4(o.i[21]) > max_ldw) max_ldw = pad4(o.i[21]);
      off_z = bump(sizeof(real) * (size_t)B * TP * max_ldw);       // per-walker pre-activation rows of the split linear layers
    }
    off_logdet = bump(sizeof(double) * (size_t)B * sys.n_det * TP);
    off_signk = bump(sizeof(int32_t) * (size_t)B * sys.n_det);
    off_cond = bump(sizeof(double) * (size_t)B * sys.n_det);      // conditioning record per determinant (Laplacian mode)
    off_kappa = bump(sizeof(double) * (size_t)B);                 // ... and its psi-weighted sum per walker
    if (ph_n && TP > 1) off_phq = bump(sizeof(double) * (size_t)B * N * dqmc::PH_STRIDE);
    if (off > ws_bytes) {
      if (d_ws) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ws)); d_ws = nullptr; ws_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ws, off);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "workspace of " + std::to_string(off) + " bytes: " + hipGetErrorString(e));
      ws_bytes = off;
    }
    last_B = B; last_TP = TP;
    return DQMC_OK;
  }
  real* bptr(int b) { return reinterpret_cast<real*>(d_ws + buf_off[b]); }

  // Workspace bytes per walker for an evaluation with TP lanes (what plan() allocates, without alignment slack).
  size_t ws_bytes_per_walker(int TP, const std::vector<char>* only = nullptr) const {
    size_t b = 0;
    for (size_t k = 0; k < bufs.size(); ++k)
      if (!only || (*only)[k]) b += sizeof(real) * (size_t)bufs[k].rows * lanes_of((int)k, TP) * bufs[k].width;
    size_t zrow = 4;
    for (const auto& o : ops) if (o.kind == DQMC_OP_LINEAR && (size_t)pad4(o.i[21]) > zrow) zrow = (size_t)pad4(o.i[21]);
    return b + sizeof(real) * zrow * TP + sizeof(double) * (size_t)sys.n_det * TP + sizeof(int32_t) * (size_t)sys.n_det +
           sizeof(double) * ((size_t)sys.n_det + 1) +           // off_cond, off_kappa
           (ph_n && TP > 1 ? sizeof(double) * (size_t)N * dqmc::PH_STRIDE : 0);
  }

  // Execute the layer program on B walkers, in chunks if the activation workspace of the whole batch would exceed
  // ws_budget (benzene/Psiformer in Laplacian mode needs ~0.2 GB per walker: 2048 walkers per GPU do not fit 288 GB
  // at once).  Chunks are still thousands of MFMA row blocks each.
  int run(const real* r, const real* R, int B, bool laplacian, real* logpsi, int32_t* sign, real* e_loc, real* stats,
          real* grad) {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    const int T = laplacian ? 3 * N + 2 : 1, TP = laplacian ? (T + 15) / 16 * 16 : 1;
    const size_t per = ws_bytes_per_walker(TP);
    long chunk = per ? (long)(ws_budget / per) : B;
    if (chunk < 1) chunk = 1;
    if (laplacian) last_chunks[0] = chunk >= B ? 1 : (int)((B + chunk - 1) / chunk);
    tail_now = false;
    if (laplacian) last_tail = false;
    if (chunk >= B) {
      if (laplacian) { tail_now = tail_ready(B, TP); last_tail = tail_now; }
      int rc;
      if (laplacian && graph_fits(B)) rc = run_graphed(r, R, B, logpsi, sign, e_loc, stats, grad);
      else rc = run_chunk(r, R, B, laplacian, logpsi, sign, e_loc, stats, B, grad, 0);
      tail_now = false;
      return rc;
    }
    for (int b0 = 0; b0 < B; b0 += (int)chunk) {
      const int nb = (B - b0) < chunk ? (B - b0) : (int)chunk;
      const int rc = run_chunk(r + (size_t)b0 * N * 3, R, nb, laplacian, logpsi ? logpsi + b0 : nullptr, sign ? sign + b0 : nullptr,
                               e_loc ? e_loc + b0 : nullptr, stats ? stats + b0 : nullptr, B, grad ? grad + (size_t)b0 * 3 * N : nullptr, b0);
      if (rc) return rc;
    }
    return DQMC_OK;
  }

  // one forward-Laplacian pass through its captured graph (see pass_graph above)
  int run_graphed(const real* r, const real* R, int B, real* logpsi, int32_t* sign, real* e_loc, real* stats, real* grad) {
    if (std::find(graph_warm.begin(), graph_warm.end(), B) == graph_warm.end()) {
      graph_warm.push_back(B);
      return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
    }
    const void* key[7] = {r, R, logpsi, sign, e_loc, stats, grad};
    PassGraph* hit = nullptr;
    for (auto& g : pgraphs)
      if (g.B == B && g.flag == flag_on && g.ws == d_ws && g.flagp == d_flag && g.epoch == graph_epoch && !memcmp(g.p, key, sizeof(key)) &&
          g.ws_tail[0] == tail_ws(0) && g.ws_tail[1] == tail_ws(1)) hit = &g;
    if (!hit) {
      // a caller that hands over different buffers on every call would pay a capture (milliseconds) per pass: once captures
      // clearly outnumber replays, the context goes back to eager launches for good
      if (graph_captures >= 8 && graph_hits < graph_captures) {
        graph_broken = true;
        drop_graphs();
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      ++graph_captures;
      if (!st_g) {
        HIP_TRY(hipStreamCreateWithFlags(&st_g, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_g0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_g1, hipEventDisableTiming));
      }
      hipStream_t caller = st;
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(st_g, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        graph_broken = true;
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      st = st_g;                                   // (run_chunk launches on `st` and forks its side streams from it)
      const int rc = run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      st = caller;
      const hipError_t ee = hipStreamEndCapture(st_g, &graph);
      hipGraphExec_t exec = nullptr;
      if (rc || ee != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        graph_broken = true;                       // this program / runtime does not capture: eager from now on
        if (rc) return rc;
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      (void)hipGraphDestroy(graph);
      if (pgraphs.size() >= 12) {                  // (a handful of batch sizes x output buffers; evict the least recently used)
        size_t old = 0;
        for (size_t k = 1; k < pgraphs.size(); ++k) if (pgraphs[k].used < pgraphs[old].used) old = k;
        if (pgraphs[old].exec) (void)hipGraphExecDestroy((hipGraphExec_t)pgraphs[old].exec);
        pgraphs.erase(pgraphs.begin() + (long)old);
      }
      PassGraph g{};
      memcpy(g.p, key, sizeof(key));
      g.B = B; g.flag = flag_on; g.ws = d_ws; g.flagp = d_flag; g.epoch = graph_epoch; g.exec = exec;
      g.ws_tail[0] = tail_ws(0); g.ws_tail[1] = tail_ws(1);      // (a captured tail holds the twin's workspace and the tail scratch)
      pgraphs.push_back(g);
      hit = &pgraphs.back();
    }
    else ++graph_hits;
    hit->used = ++graph_clock;
    HIP_TRY(hipEventRecord(ev_g0, st));
    HIP_TRY(hipStreamWaitEvent(st_g, ev_g0, 0));
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)hit->exec, st_g));
    HIP_TRY(hipEventRecord(ev_g1, st_g));
    HIP_TRY(hipStreamWaitEvent(st, ev_g1, 0));
    // the host-side layout (buf_off, off_*, last_B, last_TP) follows the replayed pass, so that dqmc_debug_read after it
    // addresses what the graph wrote; the slab is already large enough (the graph was captured on it), nothing is reallocated
    if constexpr (sizeof(real) == 4) {
      if (tail_now && twin) { const int rct = static_cast<Engine<double>*>(twin)->plan(B, (3 * N + 2 + 15) / 16 * 16, &tail_alloc); if (rct) return rct; }
    }
    return plan(B, (3 * N + 2 + 15) / 16 * 16);
  }

  int run_chunk(const real* r, const real* R, int B, bool laplacian, real* logpsi, int32_t* sign, real* e_loc, real* stats,
                long stats_ld, real* grad, int b_offset) {
    dqmc::LaneInfo li;
    li.N = N;
    li.T = laplacian ? 3 * N + 2 : 1;
    li.TP = laplacian ? (li.T + 15) / 16 * 16 : 1;
    if (!lanes_supported(li.TP)) return fail(DQMC_E_UNSUPPORTED, "no kernel instance for " + std::to_string(li.TP) + " lanes");
    int rc = plan(B, li.TP, tail_only ? &tail_alloc : nullptr);
    if (rc) return rc;
    // pseudo-Hamiltonian: the local-energy pass (not the plain gradient of psi_grad / the Langevin sampler) seeds its
    // derivative lanes with the per-electron Cholesky factors of A(r_i) (ecp/pseudo_hamiltonian.py:115-146)
    const double* phq = nullptr;
    if (laplacian && ph_n && e_loc && !ph_skip) {
      double* q = reinterpret_cast<double*>(d_ws + off_phq);
      dqmc::launch_ph_coeffs<real>(st, r, R, d_ph_nuc, ph_n, d_ph_loc, d_ph_l2, ph_grid, ph_rmax, B, N, q);
      phq = q;
    }
    // edge-stream ops (destination carries pair-compact lanes) go to the companion stream in Laplacian mode, independent
    // node branches to two more (analyse_streams); every op records an event, readers on other streams wait for the
    // events of the buffers they read
    const bool dual = laplacian && dual_stream && !timing_serial() && std::any_of(compact.begin(), compact.end(), [](char c) { return c != 0; });
    const bool multi = dual && multi_stream && (long)B * N * li.TP <= (1L << 20);      // (large batches fill the GPU kernel by kernel)
    struct BufEv { hipEvent_t ev; int sid; };
    std::vector<std::vector<BufEv>> buf_w(dual ? bufs.size() : 0);
    size_t ev_next = 0;
    hipStream_t sl[4] = {st, st, st, st};
    hipEvent_t last_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t zb_reader = nullptr;                    // last reader of the per-walker pre-activation scratch
    auto new_event = [&](hipEvent_t* out) -> int {
      if (ev_next == ms_events.size()) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ms_events.push_back(e); }
      *out = ms_events[ev_next++];
      return DQMC_OK;
    };
    if (dual) {
      if (!st2) HIP_TRY(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
      if (!ev_fork) HIP_TRY(hipEventCreate(&ev_fork));
      sl[1] = st2;
      HIP_TRY(hipEventRecord(ev_fork, st));            // inputs ready, the previous evaluation's readers done
      HIP_TRY(hipStreamWaitEvent(st2, ev_fork, 0));
      if (multi)
        for (int x = 0; x < 2; ++x) {
          if (!st_extra[x]) HIP_TRY(hipStreamCreateWithFlags(&st_extra[x], hipStreamNonBlocking));
          sl[2 + x] = st_extra[x];
          HIP_TRY(hipStreamWaitEvent(st_extra[x], ev_fork, 0));
        }
    }
    auto sid_of = [&](size_t opi) -> int {
      if (!dual) return 0;
      const int sid = op_sid[opi];
      if (sid < 0) return 0;
      return (sid >= 2 && !multi) ? 0 : sid;
    };
    std::vector<int> rd_b, wr_b;
    auto wait_buf = [&](int b, int sid) -> int {        // stream slot `sid` is about to read buffer b
      for (const BufEv& w : buf_w[b]) if (w.sid != sid) HIP_TRY(hipStreamWaitEvent(sl[sid], w.ev, 0));
      return DQMC_OK;
    };
    auto before = [&](const dqmc_op& o, int sid) -> int {
      if (!dual) return DQMC_OK;
      op_io(o, rd_b, wr_b);
      for (int b : rd_b) { const int rcw = wait_buf(b, sid); if (rcw) return rcw; }
      return DQMC_OK;
    };
    auto after = [&](const dqmc_op& o, int sid) -> int {
      if (!dual) return DQMC_OK;
      op_io(o, rd_b, wr_b);
      hipEvent_t e;
      { const int rce = new_event(&e); if (rce) return rce; }
      HIP_TRY(hipEventRecord(e, sl[sid]));
      last_ev[sid] = e;
      for (int b : wr_b) buf_w[b].push_back(BufEv{e, sid});
      return DQMC_OK;
    };
    // second layers of chained MLPs that actually ran inside their parent's launch IN THIS PASS (the chained kernel has
    // instances for some lane counts only, and needs parent and child in the same lane layout: otherwise both layers run
    // as ordinary LINEAR ops)
    std::vector<char> ran_with_parent(ops.size(), 0);
    size_t first_op = 0;
    const bool skip_tail = tail_now && laplacian;      // this context runs the head: the tail ops go to the float64 twin afterwards
    if (!laplacian && fused_n_ops > 0 && fused2_WT > 0 && (fused_enabled >= 2 || (fused_enabled == 1 && fused_pays(B)))) {
      rc = run_fused2(r, R, B, li);
      if (rc) return rc;
      first_op = (size_t)fused_n_ops;
    }
    for (size_t opi = first_op; opi < ops.size(); ++opi) {
      const dqmc_op& op = ops[opi];
      const int32_t* i = op.i;
      if (ran_with_parent[opi]) continue;
      if (skip_tail ? in_tail[opi] != 0 : (tail_only && !in_tail[opi])) continue;               // second layer of a chained MLP: ran with its parent
      const int sid = sid_of(opi);
      const hipStream_t so = sl[sid];
      { const int rcb = before(op, sid); if (rcb) return rcb; }
      switch (op.kind) {
        case DQMC_OP_FEAT_EN:
          t_begin("feat", 0, so);
          dqmc::launch_feat_en<real>(so, r, R, bptr(i[0]), B, sys.n_nuc, sys.n_up, bufs[i[0]].width, li, sys.norm_eps, i[1], i[2], phq);
          t_end();
          break;
        case DQMC_OP_FEAT_EE:
          t_begin("feat", 0, so);
          dqmc::launch_feat_ee<real>(so, r, R, d_it + i[1], bptr(i[0]), B, i[2], li, sys.norm_eps, i[3],
                                     li.TP > 1 && compact[i[0]], phq);
          t_end();
          break;
        case DQMC_OP_LINEAR: {
          dqmc::LinArgs<real> a{};
          a.cfg_bf = linear_bf; a.cfg_bkx = linear_bkx; a.cfg_f64_split = linear_f64_split;
          a.n_pieces = i[0];
          int ktot = 0, w_row = 0, n_bc = 0;
          for (int p = 0; p < i[0]; ++p) {
            const int sb = i[1 + 4 * p];
            a.piece[p].src = bptr(sb);
            a.piece[p].ld = bufs[sb].width;
            a.piece[p].rpw = bufs[sb].rows;
            a.piece[p].r0 = i[2 + 4 * p];
            a.piece[p].K = pad4(i[3 + 4 * p]);
            a.piece[p].bcast = i[4 + 4 * p];
            a.piece[p].w_row = w_row;
            w_row += pad4(i[3 + 4 * p]);
            n_bc += i[4 + 4 * p] ? 1 : 0;
            ktot += i[3 + 4 * p];
          }
          if (tail_only) {          // (the twin executing a tail: pieces the float32 context wrote are read where they lie)
            bool all32 = !src32_of.empty();
            for (int p = 0; p < i[0] && all32; ++p) all32 = src32_of[i[1 + 4 * p]] != nullptr;
            if (all32) {
              a.src_f32 = 1;
              for (int p = 0; p < i[0]; ++p) a.piece[p].src = reinterpret_cast<const real*>(src32_of[i[1 + 4 * p]]);
            }
          }
          a.W = d_w + i[22];
          a.ldw = pad4(i[21]);
          a.bias = i[23] >= 0 ? d_w + i[23] : nullptr;
          a.dst = bptr(i[17]);
          a.ld_dst = bufs[i[17]].width; a.rpw_dst = bufs[i[17]].rows; a.r0_dst = i[18]; a.col0_dst = i[19];
          a.res = i[25] >= 0 ? bptr(i[25]) : nullptr;
          if (i[25] >= 0) { a.ld_res = bufs[i[25]].width; a.rpw_res = bufs[i[25]].rows; a.r0_res = i[26]; }
          a.res_scale = i[27] ? (real)0.70710678118654752440 : (real)1;
          a.act = i[24]; a.nrows = i[20]; a.B = B; a.T = li.T; a.TP = li.TP;
          if (li.TP > 1 && compact[i[17]]) { a.T = dqmc::PAIR_LANES; a.TP = dqmc::PAIR_LANES; }   // row-wise op on edge rows
          if (mlp_child[opi] >= 0 && dqmc::linear_chain_supported(a.TP, a.ldw, pad4(ops[mlp_child[opi]].i[21])) &&
              compact[ops[mlp_child[opi]].i[17]] == compact[i[17]]) {
            // hidden layer + output layer of a row-wise MLP in one launch (the hidden activations stay in LDS)
            const dqmc_op& ch = ops[mlp_child[opi]];
            const int32_t* c = ch.i;
            { const int rcb = before(ch, sid); if (rcb) return rcb; }
            a.W2 = d_w + c[22]; a.ldw2 = pad4(c[21]); a.bias2 = c[23] >= 0 ? d_w + c[23] : nullptr; a.act2 = c[24];
            a.dst = bptr(c[17]);
            a.ld_dst = bufs[c[17]].width; a.rpw_dst = bufs[c[17]].rows; a.r0_dst = c[18]; a.col0_dst = c[19];
            a.res = c[25] >= 0 ? bptr(c[25]) : nullptr;
            if (c[25] >= 0) { a.ld_res = bufs[c[25]].width; a.rpw_res = bufs[c[25]].rows; a.r0_res = c[26]; }
            a.res_scale = c[27] ? (real)0.70710678118654752440 : (real)1;
            t_begin("linear", 2.0 * (double)B * i[20] * li.T * ((double)ktot * i[21] + (double)c[3] * c[21]), so,
                    2.0 * (double)B * i[20] * a.TP * ((double)w_row * a.ldw + (double)a.ldw * a.ldw2));
            dqmc::launch_linear_chain<real>(so, a);
            t_end();
            ran_with_parent[mlp_child[opi]] = 1;
            { const int rca = after(op, sid); if (rca) return rca; }
            { const int rca = after(ch, sid); if (rca) return rca; }
            continue;
          }
          if (split_bcast && n_bc > 0 && n_bc < i[0] && i[20] > 1 && !(li.TP > 1 && compact[i[17]])) {
            // Per-walker (broadcast) pieces -- the spin means of the node update, reference gnn/update_features.py:64-106 --
            // contribute the same row to every electron of a walker: their product with W is computed ONCE per walker into
            // a scratch row and enters the main launch as an addend of the pre-activation (for LiH 57 %, for N2 62 % of the
            // K range of the g layers; the dense path multiplied it once per electron).
            dqmc::LinArgs<real> z = a;
            dqmc::LinArgs<real> m = a;
            z.n_pieces = m.n_pieces = 0;
            for (int p = 0; p < i[0]; ++p) {
              if (a.piece[p].bcast) z.piece[z.n_pieces++] = a.piece[p];
              else m.piece[m.n_pieces++] = a.piece[p];
            }
            real* zb = reinterpret_cast<real*>(d_ws + off_z);
            z.bias = nullptr; z.act = 0; z.res = nullptr; z.pre = nullptr;
            z.dst = zb; z.ld_dst = a.ldw; z.rpw_dst = 1; z.r0_dst = 0; z.col0_dst = 0; z.nrows = 1;
            m.pre = zb; m.ld_pre = a.ldw;
            // the per-walker product runs where its inputs (the spin means) were produced, i.e. beside whatever the main
            // stream is still doing for this layer; the scratch row buffer is shared by all layers: its previous reader
            // (the last g layer) must be done before it is overwritten
            int zsid = sid;
            if (dual) {
              for (int p = 0; p < i[0]; ++p)
                if (i[4 + 4 * p] && !buf_w[i[1 + 4 * p]].empty()) zsid = buf_w[i[1 + 4 * p]].back().sid;
              for (int p = 0; p < i[0]; ++p)
                if (i[4 + 4 * p]) { const int rcw = wait_buf(i[1 + 4 * p], zsid); if (rcw) return rcw; }
              if (zb_reader && zsid != sid) HIP_TRY(hipStreamWaitEvent(sl[zsid], zb_reader, 0));
            }
            t_begin("linear", 0, sl[zsid], 2.0 * (double)B * li.TP * (double)[&] { int k = 0; for (int q = 0; q < z.n_pieces; ++q) k += z.piece[q].K; return k; }() * a.ldw);
            dqmc::launch_linear<real>(sl[zsid], z);
            t_end();
            if (dual && zsid != sid) {
              hipEvent_t ez;
              { const int rce = new_event(&ez); if (rce) return rce; }
              HIP_TRY(hipEventRecord(ez, sl[zsid]));
              last_ev[zsid] = ez;
              HIP_TRY(hipStreamWaitEvent(so, ez, 0));
            }
            t_begin("linear", 2.0 * (double)B * i[20] * li.T * (double)ktot * i[21], so,
                    2.0 * (double)B * i[20] * a.TP * (double)[&] { int k = 0; for (int q = 0; q < m.n_pieces; ++q) k += m.piece[q].K; return k; }() * a.ldw);
            dqmc::launch_linear<real>(so, m);
            t_end();
            if (dual) {
              { const int rce = new_event(&zb_reader); if (rce) return rce; }
              HIP_TRY(hipEventRecord(zb_reader, so));
            }
            break;
          }
          t_begin("linear", 2.0 * (double)B * i[20] * li.T * (double)ktot * i[21], so, 2.0 * (double)B * i[20] * a.TP * (double)w_row * a.ldw);
          dqmc::launch_linear<real>(so, a);
          t_end();
          break;
        }
        case DQMC_OP_SPIN_MEAN:
          t_begin("graph", 0, so);
          dqmc::launch_spin_mean<real>(so, bptr(i[0]), bptr(i[1]), B, i[2], bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_CONV:
          t_begin("graph", 0, so);
          dqmc::launch_conv<real>(so, bptr(i[0]), bufs[i[0]].rows, bufs[i[0]].width, bptr(i[1]), bufs[i[1]].rows, bufs[i[1]].width, bptr(i[2]),
                                  bufs[i[2]].width, i[3], d_it + i[4], i[5], i[6], B, li, li.TP > 1 && compact[i[0]]);
          t_end();
          break;
        case DQMC_OP_EDGE_SUM:
          t_begin("graph", 0, so);
          dqmc::launch_edge_sum<real>(so, bptr(i[0]), bufs[i[0]].rows, bufs[i[0]].width, bptr(i[2]), bufs[i[2]].width, i[3],
                                      d_it + i[4], i[5], i[6], 1.0 / (double)(i[1] > 0 ? i[1] : 1), B, li,
                                      li.TP > 1 && compact[i[0]]);
          t_end();
          break;
        case DQMC_OP_ATTENTION: {
          // algorithmic flops: S, dP v0 / P v_c, dP_c v_c contractions per lane
          t_begin("attention", 2.0 * B * i[4] * (double)N * (N + i[6]) * i[5] * (li.T == 1 ? 2.0 : 5.0 * li.T));
          int rc2 = DQMC_OK;
          const bool att_mfma = attention_mfma && dqmc::attention_mfma_supported<real>(N, i[5], i[6]) &&
                                (attention_mfma >= 2 || dqmc::attention_mfma_profitable(N));
          // (float64 only: the float32 instance of the split kernel agrees with float64 in the emulator but sent a whole benzene
          // batch to the float64 pass on the MI355X when it was tried -- not understood, not instantiated for the product)
          const int att_split = sizeof(real) == 8 ? (attention_split < 0 ? 1 : attention_split) : 0;
          bool done = false;
          if constexpr (sizeof(real) == 8) {
            if (att_mfma && att_split && li.TP > 1 && dqmc::attention_mfma_split_lds_bytes<real>(N, i[5], i[6]) <= (size_t)160 * 1024 &&
                i[5] >= 16) {
              rc2 = dqmc::launch_attention_mfma_split<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width, i[4], i[5],
                                                            B, li, i[6], d_w + i[7], d_w + i[8]);
              done = true;
            }
          }
          if (done) {}
          else if (att_mfma)
            rc2 = dqmc::launch_attention_mfma<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width, i[4], i[5], B, li,
                                                    i[6], d_w + i[7], d_w + i[8], attention_ncb < 0 ? (sizeof(real) == 4 ? 1 : 0) : attention_ncb);
          else
            rc2 = dqmc::launch_attention<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width,
                                               i[4], i[5], B, li, i[6], d_w + i[7], d_w + i[8]);
          t_end();
          if (rc2) return fail(DQMC_E_HIP, "attention launch failed");
          break;
        }
        case DQMC_OP_CONST:
          t_begin("feat", 0, so);
          dqmc::launch_const_rows<real>(so, d_w + i[1], bptr(i[0]), B, bufs[i[0]].rows, bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_ROW_SUM:
          t_begin("graph", 0, so);
          dqmc::launch_row_sum<real>(so, bptr(i[0]), bptr(i[1]), B, bufs[i[0]].rows, bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_ORBITALS:
          t_begin("orbitals", 0, so);
          dqmc::launch_orbitals<real>(so, r, R, bptr(i[0]), bufs[i[0]].width, bptr(i[1]), bufs[i[1]].width, d_w + i[2], d_w + i[3],
                                      d_w + i[4], d_w + i[5], B, sys.n_up, sys.n_nuc, i[6] > 0 ? i[6] : 1, sys.n_det, li,
                                      sys.norm_eps, phq);
          t_end();
          break;
        case DQMC_OP_SLOGDET:
          t_begin("slogdet", 0);
          dqmc::launch_slogdet<real>(st, bptr(i[0]), bufs[i[0]].width, reinterpret_cast<double*>(d_ws + off_logdet),
                                     reinterpret_cast<int32_t*>(d_ws + off_signk), B, sys.n_det, li, slogdet_mfma,
                                     laplacian ? reinterpret_cast<double*>(d_ws + off_cond) : nullptr);
          t_end();
          break;
        case DQMC_OP_FINAL: {
          dqmc::FinalArgs a{};
          a.r = r; a.R = R; a.charges = d_charges;
          a.ecp_loc = d_ecp_loc; a.ecp_nt = ecp_nt_loc;
          a.logdet = reinterpret_cast<double*>(d_ws + off_logdet);
          a.sign_k = reinterpret_cast<int32_t*>(d_ws + off_signk);
          a.jastrow = i[0] >= 0 ? bptr(i[0]) : nullptr;
          a.jas_width = i[0] >= 0 ? bufs[i[0]].width : 0;
          a.conf_coeff = i[1] >= 0 ? d_w + i[1] : nullptr;
          a.alphas = d_w + i[3];
          a.cusp_kind = i[2];
          a.same_scale = op.f[0]; a.anti_scale = op.f[1];
          a.eps = sys.norm_eps;
          a.B = B; a.n_up = sys.n_up; a.n_nuc = sys.n_nuc; a.K = sys.n_det; a.li = li;
          a.logpsi = logpsi; a.sign = sign; a.e_loc = e_loc; a.stats = stats; a.stats_ld = stats_ld; a.grad = grad;
          a.phq = phq;
          if (laplacian) { a.cond = reinterpret_cast<double*>(d_ws + off_cond); a.kappa_out = reinterpret_cast<double*>(d_ws + off_kappa); }
          if (flag_on && laplacian) {
            a.flag_count = d_flag; a.flag_idx = d_flag + 1; a.refine_thresh = refine_thresh; a.thresh_dev = d_thresh; a.b_offset = b_offset;
            a.score_out = d_score;
          }
          t_begin("final", 0);
          dqmc::launch_final<real>(st, a);
          t_end();
          break;
        }
        default:
          return fail(DQMC_E_UNSUPPORTED, "op kind " + std::to_string(op.kind));
      }
      { const int rca = after(op, sid); if (rca) return rca; }
    }
    if (dual)        // join: nothing of this evaluation may still run on another stream when the caller goes on
      for (int x = 1; x < 4; ++x) if (last_ev[x]) HIP_TRY(hipStreamWaitEvent(st, last_ev[x], 0));
    HIP_TRY(hipGetLastError());
    if (tail_now && laplacian) return run_tail(r, R, B, logpsi, sign, e_loc, stats, stats_ld, grad, b_offset);
    return DQMC_OK;
  }
  bool timing_serial() const { return false; }
  // The LDS-resident kernel is the faster VALUE path for the small systems it was built for (N <= 4: one launch per
  // Metropolis sub-step) and for small batches of any system (one launch against ~50); for larger systems at large
  // batch the layered MFMA kernels win (N2 / FermiNet, 4096 walkers: 1.38 ms against 1.61 ms per value pass).
  bool fused_pays(int B) const { return N <= 4 || B < 1024; }

  int wf_eval(const void* r, const void* R, int B, void* logpsi, int32_t* sign) override {
    return run((const real*)r, (const real*)R, B, false, (real*)logpsi, sign, nullptr, nullptr, nullptr);
  }
  int local_energy(const void* r, const void* R, int B, void* e_loc, void* stats, void* grad, void* logpsi,
                   int32_t* sign) override {
    last_chunks[0] = last_chunks[1] = 0;
    return lap_refined((const real*)r, (const real*)R, B, (real*)e_loc, (real*)stats, (real*)grad, (real*)logpsi, sign);
  }
  // the pass in this context's own precision: forward-Laplacian evaluation, plus the non-local ECP quadrature when the
  // Hamiltonian has one and a local energy is asked for
  int pass_own(const real* r, const real* R, int B, real* logpsi, int32_t* sign, real* e_loc, real* stats, real* grad) {
    if (ecp_n_nl == 0 || !e_loc || ecp_skip_nl || ecp_defer) return run(r, R, B, true, logpsi, sign, e_loc, stats, grad);
    return local_energy_ecp(r, R, B, e_loc, stats, grad, logpsi, sign);
  }

  // log|psi|, sign and grad log|psi| from the forward-Laplacian pass alone: no potentials beyond k_final's, no
  // non-local ECP quadrature (what value_and_grad(psi) gives the reference's Langevin sampler)
  int psi_grad(const void* r, const void* R, int B, void* logpsi, int32_t* sign, void* grad) override {
    return lap_refined((const real*)r, (const real*)R, B, nullptr, nullptr, (real*)grad, (real*)logpsi, sign);
  }

  // The forward-Laplacian pass; in the float32 build followed by the float64 re-evaluation of the walkers k_final
  // flagged (a few per cent of |psi|^2-distributed walkers of a small system; option "refine" 0 turns it off).
