// engine_fused_plan.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// planner of the LDS-resident value kernel (kernel_fused2.hip): dependency levels, LDS layout, unit descriptors per wave,
// fragment-major weight packing, the launch.

  static void op_io(const dqmc_op& op, std::vector<int>& rd, std::vector<int>& wr) {
    const int32_t* i = op.i;
    rd.clear(); wr.clear();
    switch (op.kind) {
      case DQMC_OP_FEAT_EN: case DQMC_OP_FEAT_EE: case DQMC_OP_CONST: wr.push_back(i[0]); break;
      case DQMC_OP_LINEAR:
        for (int p = 0; p < i[0]; ++p) rd.push_back(i[1 + 4 * p]);
        if (i[25] >= 0) rd.push_back(i[25]);
        wr.push_back(i[17]); break;
      case DQMC_OP_SPIN_MEAN: case DQMC_OP_ROW_SUM: rd.push_back(i[0]); wr.push_back(i[1]); break;
      case DQMC_OP_CONV: rd.push_back(i[0]); rd.push_back(i[1]); wr.push_back(i[2]); break;
      case DQMC_OP_EDGE_SUM: rd.push_back(i[0]); wr.push_back(i[2]); break;
      case DQMC_OP_ORBITALS: rd.push_back(i[0]); wr.push_back(i[1]); break;
      case DQMC_OP_SLOGDET: rd.push_back(i[0]); break;
      case DQMC_OP_FINAL: if (i[0] >= 0) rd.push_back(i[0]); break;
      default: break;
    }
  }

  // Dependency levels of ops [0, fused_n_ops): ops of one level are independent and run
  // without a workgroup barrier between them inside the fused kernel.
  void fused_schedule() {
    const int no = fused_n_ops, nb = (int)bufs.size();
    std::vector<int> lvl(no, 0), wlevel(nb, -1);
    std::vector<int> rd, wr;
    if (fused_sched_mode == 0) {
      // Program order kept (so LDS liveness is what the program compiler laid out); a new level
      // starts whenever an op reads a buffer written inside the current level.
      int cur = 0;
      std::vector<char> written(nb, 0);
      f_order.resize(no);
      f_level.assign(no, 0);
      for (int k = 0; k < no; ++k) {
        op_io(ops[k], rd, wr);
        bool dep = false;
        for (int b : rd) dep = dep || written[b];
        if (dep) { ++cur; std::fill(written.begin(), written.end(), 0); }
        for (int b : wr) written[b] = 1;
        f_order[k] = k;
        f_level[k] = cur;
      }
      return;
    }
    for (int k = 0; k < no; ++k) {
      op_io(ops[k], rd, wr);
      int l = 0;
      for (int b : rd) if (wlevel[b] + 1 > l) l = wlevel[b] + 1;
      for (int b : wr) if (wlevel[b] >= 0 && wlevel[b] > l) l = wlevel[b];   // co-writers share a level or later
      lvl[k] = l;
      for (int b : wr) if (l > wlevel[b]) wlevel[b] = l;
    }
    // a buffer's readers must come after ALL its writers: raise readers to max writer level + 1
    for (bool changed = true; changed;) {
      changed = false;
      std::fill(wlevel.begin(), wlevel.end(), -1);
      for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) if (lvl[k] > wlevel[b]) wlevel[b] = lvl[k]; }
      for (int k = 0; k < no; ++k) {
        op_io(ops[k], rd, wr);
        for (int b : rd) if (wlevel[b] >= 0 && lvl[k] <= wlevel[b]) { lvl[k] = wlevel[b] + 1; changed = true; }
      }
    }
    if (fused_sched_mode == 3) {
      // List scheduling under an LDS budget: level by level, ops on the critical path (no slack against the
      // as-late-as-possible levels) are placed unconditionally, the others -- in program order -- only while
      // the bytes live in the level stay under the budget.  Gives (nearly) the short critical path of the full
      // levels with (nearly) the footprint of program order, i.e. one more co-resident workgroup per CU.
      int L = 0;
      for (int k = 0; k < no; ++k) L = lvl[k] + 1 > L ? lvl[k] + 1 : L;
      std::vector<int> alap(no, L - 1), rlevel(nb, L);
      for (int k = no - 1; k >= 0; --k) {
        op_io(ops[k], rd, wr);
        int l = L - 1;
        for (int b : wr) if (rlevel[b] - 1 < l) l = rlevel[b] - 1;
        if (l < lvl[k]) l = lvl[k];
        alap[k] = l;
        for (int b : rd) if (l < rlevel[b]) rlevel[b] = l;
      }
      // per buffer: writers, readers; global buffers (read after the fused range) cost no LDS
      std::vector<std::vector<int>> writers(nb), readers(nb);
      for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) writers[b].push_back(k); for (int b : rd) readers[b].push_back(k); }
      std::vector<char> is_glob(nb, 0);
      for (int k = no; k < (int)ops.size(); ++k) { op_io(ops[k], rd, wr); for (int b : rd) is_glob[b] = 1; }
      const int WTl = fused_sched_wt > 0 ? fused_sched_wt : 4;
      auto blen = [&](int b) { return is_glob[b] ? (size_t)0 : sizeof(real) * (((size_t)WTl * bufs[b].rows * (bufs[b].width + 2) + 3) / 4 * 4); };
      std::vector<int> sched(no, -1);
      int n_done = 0, delay = 0;
      for (int l = 0; n_done < no; ++l) {
        auto ready = [&](int k) {
          op_io(ops[k], rd, wr);
          for (int b : rd) for (int w : writers[b]) if (sched[w] < 0 || sched[w] >= l) return false;
          return true;
        };
        auto live_bytes = [&]() {       // buffers with a scheduled writer and a reader not scheduled before level l
          size_t tot = 0;
          for (int b = 0; b < nb; ++b) {
            bool written = false, needed = false;
            for (int w : writers[b]) written = written || sched[w] >= 0;
            if (!written) continue;
            for (int rr : readers[b]) needed = needed || sched[rr] < 0 || sched[rr] >= l;
            for (int w : writers[b]) needed = needed || sched[w] == l || sched[w] < 0;
            if (needed) tot += blen(b);
          }
          return tot;
        };
        std::vector<int> cand;
        for (int k = 0; k < no; ++k) if (sched[k] < 0 && ready(k)) cand.push_back(k);
        int placed_now = 0;
        std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return alap[x] < alap[y]; });   // least slack first
        for (int k : cand) {
          sched[k] = l;
          if (live_bytes() <= fused_sched_budget) { ++n_done; ++placed_now; }
          else sched[k] = -1;
        }
        if (placed_now == 0 && !cand.empty()) { sched[cand[0]] = l; ++n_done; }   // budget too small: make progress
        (void)delay;
      }
      lvl = sched;
    }
    if (fused_sched_mode == 2) {
      // As late as possible within the same number of levels: ops with slack (the edge-stream MLPs, which do
      // not depend on the node stream) move next to their consumers, which shortens buffer live ranges and
      // so the LDS footprint of a tile (what decides how many workgroups share a CU).
      int L = 0;
      for (int k = 0; k < no; ++k) L = lvl[k] + 1 > L ? lvl[k] + 1 : L;
      std::vector<int> alap(no, L - 1), rlevel(nb, L);     // rlevel[b]: earliest level of a reader of b
      for (int k = no - 1; k >= 0; --k) {
        op_io(ops[k], rd, wr);
        int l = L - 1;
        for (int b : wr) if (rlevel[b] - 1 < l) l = rlevel[b] - 1;
        if (l < lvl[k]) l = lvl[k];
        alap[k] = l;
        for (int b : rd) if (l < rlevel[b]) rlevel[b] = l;
      }
      lvl = alap;
    }
    f_order.resize(no);
    for (int k = 0; k < no; ++k) f_order[k] = k;
    // (within a level the program order is kept)
    std::stable_sort(f_order.begin(), f_order.end(), [&](int x, int y) { return lvl[x] < lvl[y]; });
    f_level.assign(no, 0);
    for (int j = 0; j < no; ++j) f_level[j] = lvl[f_order[j]];
  }

  // LDS placement of the buffers for a tile of WT walkers: interval colouring over live level ranges.
  size_t fused_layout(int WT, std::vector<dqmc::FusedBuf>& fb) const {
    const int nb = (int)bufs.size(), no = fused_n_ops;
    const int BIG = 1 << 30;
    std::vector<int> first(nb, BIG), last(nb, -1);
    std::vector<int> rd, wr;
    for (int j = 0; j < no; ++j) {
      op_io(ops[f_order[j]], rd, wr);
      for (int b : wr) { if (f_level[j] < first[b]) first[b] = f_level[j]; if (f_level[j] > last[b]) last[b] = f_level[j]; }
      for (int b : rd) { if (f_level[j] < first[b]) first[b] = f_level[j]; if (f_level[j] > last[b]) last[b] = f_level[j]; }
    }
    for (int k = no; k < (int)ops.size(); ++k) {      // consumers after the fused range: keep in HBM
      op_io(ops[k], rd, wr);
      for (int b : rd) last[b] = BIG;
    }
    fb.assign(nb, dqmc::FusedBuf{});
    // Placement = interval colouring: buffers in decreasing size, each at the lowest offset that does not
    // overlap an already placed buffer whose live range [first, last] intersects its own (largest-first beats
    // first-fit-in-time by ~10 % here, which decides how many workgroups share a CU).
    struct Seg { size_t off, len; int a, b; };
    std::vector<Seg> placed;
    const size_t base = 0;
    size_t peak = base;
    std::vector<int> order;
    for (int b = 0; b < nb; ++b) {
      if (first[b] == BIG) continue;            // not touched by the fused range
      dqmc::FusedBuf& f = fb[b];
      f.rows = bufs[b].rows; f.width = bufs[b].width;
      if (last[b] == BIG) { f.is_global = 1; continue; }
      f.is_global = 0;
      f.stride = bufs[b].width + 2;
      order.push_back(b);
    }
    auto len_of = [&](int b) { return ((size_t)WT * fb[b].rows * fb[b].stride + 3) / 4 * 4; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len_of(x) > len_of(y); });
    if (getenv("DQMC_FUSED_VERBOSE")) {
      const int n_levels = no ? f_level[no - 1] + 1 : 0;
      for (int l = 0; l < n_levels; ++l) {
        size_t live = 0;
        std::string who;
        for (int b : order) if (first[b] <= l && l <= last[b]) { live += len_of(b); who += " " + std::to_string(b) + ":" + std::to_string(len_of(b) * sizeof(real)); }
        fprintf(stderr, "[dqmc] WT=%d level %d live %zu B:%s\n", WT, l, live * sizeof(real), who.c_str());
      }
    }
    for (int b : order) {
      const size_t len = len_of(b);
      size_t off = base;
      for (bool moved = true; moved;) {
        moved = false;
        for (const Seg& s : placed)
          if (s.a <= last[b] && first[b] <= s.b && off < s.off + s.len && s.off < off + len) { off = s.off + s.len; moved = true; }
      }
      fb[b].off = (int)off;
      placed.push_back(Seg{off, len, first[b], last[b]});
      if (off + len > peak) peak = off + len;
    }
    return peak * sizeof(real);
  }

  int build_fused_plan() {
    fused_n_ops = 0;
    int n_f = -1;
    for (int k = 0; k < (int)ops.size(); ++k) {
      if (ops[k].kind == DQMC_OP_ORBITALS) { n_f = k + 1; break; }
      if (ops[k].kind == DQMC_OP_ATTENTION || ops[k].kind == DQMC_OP_SLOGDET || ops[k].kind == DQMC_OP_FINAL) return DQMC_OK;
    }
    if (n_f < 0) return DQMC_OK;
    for (int k = 0; k < n_f; ++k) {       // the LDS-resident kernel implements tanh / silu layers over electron senders only
      if (ops[k].kind == DQMC_OP_CONST) return DQMC_OK;
      if (ops[k].kind == DQMC_OP_LINEAR && ops[k].i[24] > 2) return DQMC_OK;
      if (ops[k].kind == DQMC_OP_FEAT_EE)
        for (int r = 0; r < ops[k].i[2]; ++r) if (h_itable[ops[k].i[1] + 2 * r + 1] < 0) return DQMC_OK;
    }
    fused_n_ops = n_f;
    if (fused_sched_mode == 3) {
      // largest per-level budget (= fewest levels) whose packed 4-walker tile leaves room for 4 workgroups per CU
      bool fit = false;
      for (int kb = 44; kb >= 24 && !fit; --kb) {
        fused_sched_budget = (size_t)kb * 1024;
        fused_schedule();
        std::vector<dqmc::FusedBuf> fbt;
        fit = fused_layout(4, fbt) + 16 + dqmc::fused2_scratch_bytes(4, N, sys.n_det, (int)sizeof(real), (int)n_itable) <= fused2_lds_quarter;
      }
      if (!fit) { fused_sched_mode = 1; fused_schedule(); fused_sched_mode = 3; }   // too big for that: full levels
    } else {
      fused_schedule();
    }
    if (!d_ops) {
      HIP_TRY(hipMalloc((void**)&d_ops, sizeof(dqmc_op) * ops.size()));
      HIP_TRY(hipMalloc((void**)&d_wpk_off, 2 * sizeof(int32_t) * ops.size()));
    }
    std::vector<dqmc_op> sched(fused_n_ops);
    for (int j = 0; j < fused_n_ops; ++j) sched[j] = ops[f_order[j]];
    HIP_TRY(hipMemcpy(d_ops, sched.data(), sizeof(dqmc_op) * fused_n_ops, hipMemcpyHostToDevice));
    const int rc = pack_fused_weights();
    if (rc) return rc;
    return build_fused2_plan();
  }

  // Work lists of the descriptor-driven fused kernel (kernel_fused2.hip): the units of every linear layer,
  // dealt to the 4 waves level by level (longest first onto the least loaded wave), structured ops for all
  // waves, one barrier per dependency level.  Tile layout [row][WT] needs a power-of-two tile.
  int build_fused2_plan() {
    fused2_WT = 0;
    if (fused_n_ops == 0) return DQMC_OK;
    const int cand[] = {16, 8, 4, 2, 1};
    std::vector<dqmc::FusedBuf> fb;
    auto with_scratch = [&](size_t act, int WT) { return (act + 15) / 16 * 16 + (size_t)dqmc::fused2_scratch_bytes(WT, N, sys.n_det, (int)sizeof(real), (int)n_itable); };
    if (fused_wt_req <= 0 && with_scratch(fused_layout(4, fb), 4) <= fused2_lds_quarter) {
      // 4 walkers per tile and 4 tiles per CU: for the batch sizes of the north star (4096 walkers = 1024 tiles =
      // 256 CUs x 4) the whole batch is ONE round of co-resident workgroups (measured fastest)
      fused2_WT = 4; fused2_lds = with_scratch(fused_layout(4, fb), 4);
    } else {
      for (int WT : cand) {
        if (fused_wt_req > 0 && WT != fused_wt_req) continue;
        const size_t bytes = with_scratch(fused_layout(WT, fb), WT);
        if (bytes <= (fused_wt_req > 0 ? (size_t)160 * 1024 : fused2_lds_budget)) { fused2_WT = WT; fused2_lds = bytes; break; }
      }
    }
    if (fused2_WT == 0) return DQMC_OK;
    const int WT = fused2_WT, n_waves = 4;
    fused2_shift = 0;
    while ((1 << fused2_shift) < WT) ++fused2_shift;
    std::vector<int32_t> words(2 * (size_t)fused_n_ops);
    HIP_TRY(hipMemcpy(words.data(), d_wpk_off, sizeof(int32_t) * words.size(), hipMemcpyDeviceToHost));
    std::vector<std::vector<dqmc::FDesc>> lists(n_waves);
    struct Unit { dqmc::FDesc d; long cost; int opidx; };
    std::vector<Unit> level_units;
    std::vector<int> level_generic;
    auto flush_level = [&]() {
      // units go longest first onto the least loaded wave
      std::vector<Unit> jobs(level_units);
      std::stable_sort(jobs.begin(), jobs.end(), [](const Unit& x, const Unit& y) { return x.cost > y.cost; });
      long load[4] = {0, 0, 0, 0};
      for (const Unit& jb : jobs) {
        int best = 0;
        for (int w = 1; w < n_waves; ++w) if (load[w] < load[best]) best = w;
        lists[best].push_back(jb.d);
        load[best] += jb.cost;
      }
      for (int j : level_generic)
        for (int w = 0; w < n_waves; ++w) { dqmc::FDesc g{}; g.kind = 3; g.op = j; lists[w].push_back(g); }
      for (int w = 0; w < n_waves; ++w) { dqmc::FDesc b{}; b.kind = 2; lists[w].push_back(b); }
      level_units.clear(); level_generic.clear();
    };
    for (int j = 0; j < fused_n_ops; ++j) {
      const dqmc_op& op = ops[f_order[j]];
      const int32_t* i = op.i;
      if (op.kind != DQMC_OP_LINEAR) {
        level_generic.push_back(j);
      } else {
        const int ldw = pad4(i[21]), Rtot = WT * i[20];
        const int NRB = (Rtot + 15) / 16, NCB = (ldw + 15) / 16, n_cg = (NCB + 1) / 2;
        int rpu = NRB * n_cg / n_waves;
        rpu = rpu < 1 ? 1 : (rpu > 4 ? 4 : rpu);
        dqmc::FDesc t{};
        t.kind = 1; t.op = j; t.n_pieces = i[0]; t.rtot = Rtot; t.ldw = ldw;
        long kq = 0;
        bool bf = op_bf[j] != 0;
        for (int p = 0; p < i[0]; ++p) {
          const dqmc::FusedBuf& sb = fb[i[1 + 4 * p]];
          if (sb.is_global) { fused2_WT = 0; return DQMC_OK; }
          t.a_base[p] = sb.off + i[2 + 4 * p] * WT * sb.stride;
          t.a_stride[p] = sb.stride;
          if ((t.a_base[p] | t.a_stride[p]) & 1) bf = false;        // (8-byte LDS reads of the A octets)
        }
        if (op_bf[j] && !bf) return fail(DQMC_E_UNSUPPORTED, "fused plan: odd LDS offset under a bf16-packed layer");
        for (int p = 0; p < i[0]; ++p) {
          if (bf) {       // bf16 matrix pipe: octets of k, chunks of 32 k
            t.a_ks[p] = i[3 + 4 * p] / 8;
            t.a_nq[p] = (i[3 + 4 * p] + 31) / 32;
          } else {
            t.a_ks[p] = pad4(i[3 + 4 * p]) / 4;
            t.a_nq[p] = (t.a_ks[p] + 3) / 4;
          }
          if (i[4 + 4 * p]) t.bcast |= 1 << p;
          kq += t.a_nq[p];
        }
        t.qstride = bf ? NCB * 192 : NCB * 64;
        t.bias_off = i[23];
        const dqmc::FusedBuf& db = fb[i[17]];
        t.flags = (i[24] & 3) | (i[27] ? 4 : 0);
        if (db.is_global) { t.flags |= 8; t.dst_base = i[17]; t.g_r0 = i[18]; t.g_col0 = i[19]; }
        else { t.dst_base = db.off + i[18] * WT * db.stride + i[19]; t.dst_stride = db.stride; }
        t.res_base = -1;
        if (i[25] >= 0) {
          const dqmc::FusedBuf& rb = fb[i[25]];
          if (rb.is_global) { fused2_WT = 0; return DQMC_OK; }
          t.res_base = rb.off + i[26] * WT * rb.stride + i[19];
          t.res_stride = rb.stride;
        }
        for (int rb0 = 0; rb0 < NRB; rb0 += rpu)
          for (int cg = 0; cg < n_cg; ++cg) {
            Unit u{t, 0, f_order[j]};
            u.d.ma = (NRB - rb0) < rpu ? (NRB - rb0) : rpu;
            u.d.row0 = rb0 * 16;
            u.d.col0 = cg * 32;
            u.d.w_off = words[2 * j] / 4 + (cg * 2) * (bf ? 192 : 64);
            u.d.w_cb1 = (cg * 2 + 1 < NCB) ? (bf ? 192 : 64) : 0;
            if ((rb0 + u.d.ma) * 16 <= Rtot && cg * 32 + 32 <= ldw) u.d.flags |= 16;
            if (bf) u.d.kind = 6;
            // small layers take the lean unit body (kernel_fused2.hip: fused2_unit_lean / FusedBfUnit::lean)
            if (fused_lean && u.d.ma == 1 && i[0] == 1 && !u.d.bcast && t.a_nq[0] <= (bf ? 1 : dqmc::FusedGroup<real>::P) && !(t.flags & 8) && (i[24] & 3) <= 1) u.d.kind = bf ? 7 : 5;
            u.cost = 12 + (long)u.d.ma * ((bf ? 3 : 4) * kq + 6);     // ~ fixed setup + MFMA quads / chunks + epilogue, in 100-cycle units
            level_units.push_back(u);
          }
      }
      if (j + 1 == fused_n_ops || f_level[j + 1] > f_level[j]) flush_level();
    }
    fused2_ma1 = true;
    auto is_unit = [](int k) { return k == 1 || k == 5 || k == 6 || k == 7; };
    for (auto& l : lists) for (auto& dd : l) if (is_unit(dd.kind) && dd.ma != 1) fused2_ma1 = false;
    std::vector<dqmc::FDesc> flat;
    int32_t begin[8];
    // (plan_lists is recorded after the chaining below)
    for (int w = 0; w < n_waves; ++w) {
      begin[w] = (int32_t)flat.size();
      begin[4 + w] = -1;
      int last_unit = -1;                          // chain the units of the list: each prefetches the next one's first weights
      for (size_t k = 0; k < lists[w].size(); ++k) {
        if (!is_unit(lists[w][k].kind)) continue;
        if (last_unit < 0) begin[4 + w] = begin[w] + (int32_t)k;
        last_unit = (int)k;
      }
      last_unit = -1;
      for (size_t k = lists[w].size(); k-- > 0;) {     // every unit carries the first-group parameters of the unit after it
        dqmc::FDesc& u = lists[w][k];
        if (!is_unit(u.kind)) continue;
        const dqmc::FDesc& nx = last_unit < 0 ? u : lists[w][last_unit];
        const bool nx_bf = nx.kind == 6 || nx.kind == 7;          // (its first group: the three planes of its first chunk)
        u.nx_w_off = nx.w_off; u.nx_cb1 = nx.w_cb1; u.nx_qstride = nx_bf ? 64 : nx.qstride; u.nx_nq = nx_bf ? 3 : nx.a_nq[0];
        last_unit = (int)k;
      }
      flat.insert(flat.end(), lists[w].begin(), lists[w].end());
      dqmc::FDesc e{}; e.kind = 0; flat.push_back(e);
    }
    plan_lists = lists;
    if (d_descs) { HIP_TRY(hipFree(d_descs)); d_descs = nullptr; }
    HIP_TRY(hipMalloc((void**)&d_descs, sizeof(dqmc::FDesc) * flat.size()));
    if (!d_wave_begin) HIP_TRY(hipMalloc((void**)&d_wave_begin, sizeof(int32_t) * 8));
    if (!d_fbufs2) HIP_TRY(hipMalloc((void**)&d_fbufs2, sizeof(dqmc::FusedBuf) * bufs.size()));
    HIP_TRY(hipMemcpy(d_descs, flat.data(), sizeof(dqmc::FDesc) * flat.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wave_begin, begin, sizeof(begin), hipMemcpyHostToDevice));
    fbufs2_h = fb;
    fbufs2_uploaded = false;
    if (dqmc::fused2_set_lds_limit<real>(fused2_lds) != 0) { fused2_WT = 0; return DQMC_OK; }
    return DQMC_OK;
  }

  // LDS offset for the Slater matrices of the sub-step tail: dead activation space clear of the backflow
  // buffer, which the tail still reads; -1 if there is none (then the staged path runs).
  int substep_mat_off() const {
    int orb = -1;
    for (int j = 0; j < fused_n_ops; ++j) if (ops[f_order[j]].kind == DQMC_OP_ORBITALS) orb = f_order[j];
    if (orb < 0 || fused2_WT == 0) return -1;
    const dqmc::FusedBuf& bfb = fbufs2_h[ops[orb].i[0]];
    const size_t need = (size_t)fused2_WT * sys.n_det * N * N;
    const size_t bf_len = (size_t)fused2_WT * bfb.rows * bfb.stride;
    const size_t act_end = (fused2_lds - (size_t)dqmc::fused2_scratch_bytes(fused2_WT, N, sys.n_det, (int)sizeof(real), (int)n_itable)) / sizeof(real);
    if (bfb.is_global) return -1;
    if ((size_t)bfb.off >= need) return 0;
    if ((size_t)bfb.off + bf_len + need <= act_end) return (int)((size_t)bfb.off + bf_len);
    return -1;
  }

  int run_fused2(const real* r, const real* R, int B, dqmc::LaneInfo li, const dqmc::FusedMc* mc = nullptr) {
    bool changed = !fbufs2_uploaded;       // the buffer table goes to the device only when an offset moved (not once per sub-step)
    for (size_t b = 0; b < bufs.size(); ++b) {
      if (fbufs2_h[b].goff != (long)buf_off[b]) changed = true;
      fbufs2_h[b].goff = (long)buf_off[b];
    }
    if (changed) {
      HIP_TRY(hipMemcpyAsync(d_fbufs2, fbufs2_h.data(), sizeof(dqmc::FusedBuf) * bufs.size(), hipMemcpyHostToDevice, st));
      if (!fbufs2_uploaded) HIP_TRY(hipStreamSynchronize(st));   // (the host vector may change before an asynchronous copy from pageable memory ran)
      fbufs2_uploaded = true;
    }
    dqmc::Fused2Args<real> a{};
    a.descs = d_descs; a.wave_begin = d_wave_begin; a.ops = d_ops; a.fbufs = d_fbufs2;
    a.w = d_w; a.wpk = d_wpk; a.itable = d_it; a.ws = d_ws; a.r = r; a.R = R;
    a.B = B; a.WT = fused2_WT; a.wt_shift = fused2_shift; a.n_up = sys.n_up; a.n_nuc = sys.n_nuc; a.K = sys.n_det;
    a.li = li; a.eps = sys.norm_eps; a.prof = fused_dbg ? d_prof : nullptr;
    a.prof_wg = (fused_dbg & 2) ? d_prof + 9 * ops.size() + 80 + 1024 : nullptr;
    a.scratch_off = (int)((fused2_lds - (size_t)dqmc::fused2_scratch_bytes(fused2_WT, N, sys.n_det, (int)sizeof(real), (int)n_itable)) / sizeof(real));
    a.it_off = (int)(fused2_lds - (size_t)((4 * n_itable + 15) / 16 * 16));
    a.n_it = (int)n_itable;
    a.ma1 = fused2_ma1 ? 1 : 0;
    a.stagger_div = fused_stagger_div; a.prio_mode = fused_prio;
    if (mc) a.mc = *mc;
    double flops = 0;
    for (int k = 0; k < fused_n_ops; ++k)
      if (ops[k].kind == DQMC_OP_LINEAR) {
        int ktot = 0;
        for (int p = 0; p < ops[k].i[0]; ++p) ktot += ops[k].i[3 + 4 * p];
        flops += 2.0 * B * ops[k].i[20] * (double)ktot * ops[k].i[21];
      }
    t_begin(mc ? "fused_substep" : "fused_psi", flops);
    int occ = fused_occ_req;
    if (occ <= 0) { const size_t per_cu = (size_t)160 * 1024 / (fused2_lds ? fused2_lds : 1); occ = per_cu >= 4 ? 4 : (per_cu >= 3 ? 3 : 2); }
    dqmc::launch_fused2_value<real>(st, a, (B + fused2_WT - 1) / fused2_WT, fused2_lds, occ);
    t_end();
    return DQMC_OK;
  }

  // piece pl (0..2) of the three-bf16 split of a float (round to nearest even, residuals exact: common.h bf_split8)
  static uint16_t bf16_piece(float v, int pl) {
    auto rne = [](float f) -> uint16_t {
      uint32_t u; memcpy(&u, &f, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
      return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    auto up = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    uint16_t h = rne(v);
    for (int k = 0; k < pl; ++k) { v = v - up(h); h = rne(v); }
    return h;
  }
  // Fragment-major copy of the Linear weights: [k/4][column block][lane] so that one wave load
  // of 64 consecutive elements is exactly the MFMA B operand (B[k = l>>4][col = l&15]).
  // Per scheduled op two words go to the device: packed-weight offset and "barrier after".
  int pack_fused_weights() {
    std::vector<int32_t> words(2 * (size_t)fused_n_ops, 0);
    std::vector<real> pk;
    op_bf.assign((size_t)fused_n_ops, 0);
    for (int j = 0; j < fused_n_ops && fused_bf && sizeof(real) == 4; ++j) {
      // a layer goes to the bf16 matrix pipe when every piece is whole octets wide and the chunks of 32 k
      // (12 MFMAs of 16 cycles + the operand split) cost less than its k-steps of 4 (2 MFMAs of 32 cycles)
      const dqmc_op& op = ops[f_order[j]];
      if (op.kind != DQMC_OP_LINEAR) continue;
      bool ok = true;
      int chunks = 0, ksteps = 0;
      for (int p = 0; p < op.i[0]; ++p) {
        const int K = op.i[3 + 4 * p];
        if (K % 8 != 0 || K == 0 || (bufs[op.i[1 + 4 * p]].width & 1)) ok = false;      // (even LDS row stride: 8-byte reads of the A octets)
        chunks += (K + 31) / 32; ksteps += (K + 3) / 4;
      }
      op_bf[j] = ok && (fused_bf >= 2 ? 9 * chunks < 2 * ksteps : 4 * chunks <= ksteps);     // (option value 2: the stricter rule; measured 110.8 vs 108.8 us)
    }
    for (int j = 0; j < fused_n_ops; ++j) {
      const dqmc_op& op = ops[f_order[j]];
      words[2 * j + 1] = (j + 1 == fused_n_ops || f_level[j + 1] > f_level[j]) ? 1 : 0;
      if (op.kind != DQMC_OP_LINEAR) continue;
      const int32_t* i = op.i;
      const int ldw = pad4(i[21]), NCB = (ldw + 15) / 16;
      // quad-interleaved: [piece][quad of 4 k-steps][column block][lane][k-step in quad]; every
      // piece is zero padded to whole quads, so one 16-byte load per lane feeds 4 MFMA k-steps
      words[2 * j] = (int32_t)pk.size();
      const real* W = wtmp.data() + i[22];
      int row0 = 0;
      if (op_bf[j]) {
        // bf16 plane layout: [piece][chunk of 32 k][column block][plane][lane][4 words]; word jj of lane l holds
        // k = 32 c + 8 (l >> 4) + 2 jj (low half) and + 1 (high half) of column cb 16 + (l & 15)
        for (int p = 0; p < i[0]; ++p) {
          const int K = i[3 + 4 * p], NC = (K + 31) / 32;
          for (int c = 0; c < NC; ++c)
            for (int cb = 0; cb < NCB; ++cb)
              for (int pl = 0; pl < 3; ++pl)
                for (int l = 0; l < 64; ++l)
                  for (int jj = 0; jj < 4; ++jj) {
                    uint32_t word = 0;
                    for (int h = 0; h < 2; ++h) {
                      const int k = 32 * c + 8 * (l >> 4) + 2 * jj + h, col = cb * 16 + (l & 15);
                      const float wv = (k < K && col < ldw) ? (float)W[(size_t)(row0 + k) * ldw + col] : 0.0f;
                      word |= (uint32_t)bf16_piece(wv, pl) << (16 * h);
                    }
                    real as_real;
                    memcpy(&as_real, &word, 4);       // (float engines only: sizeof(real) == 4)
                    pk.push_back(as_real);
                  }
          row0 += K;
        }
        continue;
      }
      for (int p = 0; p < i[0]; ++p) {
        const int KS = pad4(i[3 + 4 * p]) / 4, NQ = (KS + 3) / 4;
        for (int q = 0; q < NQ; ++q)
          for (int cb = 0; cb < NCB; ++cb)
            for (int l = 0; l < 64; ++l)
              for (int jj = 0; jj < 4; ++jj) {
                const int ks = q * 4 + jj, row = row0 + ks * 4 + (l >> 4), col = cb * 16 + (l & 15);
                pk.push_back((ks < KS && col < ldw) ? W[(size_t)row * ldw + col] : (real)0);
              }
        row0 += KS * 4;
      }
    }
    if (pk.size() > wpk_cap) {
      if (d_wpk) HIP_TRY(hipFree(d_wpk));
      HIP_TRY(hipMalloc((void**)&d_wpk, sizeof(real) * pk.size()));
      wpk_cap = pk.size();
    }
    HIP_TRY(hipMemcpyAsync(d_wpk, pk.data(), sizeof(real) * pk.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_wpk_off, words.data(), sizeof(int32_t) * words.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }
