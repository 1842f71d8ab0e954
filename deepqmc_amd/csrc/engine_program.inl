// engine_program.inl -- member functions of Engine<real>, included INSIDE the struct body by engine.hip (one translation unit):
// the layer program on the host: validation (Engine::validate), chained-MLP / stream / pair-compact-lane analyses,
// weight upload, the option table.

  int validate() {
    const int nb = (int)bufs.size();
    auto okb = [&](int b) { return b >= 0 && b < nb; };
    for (size_t k = 0; k < ops.size(); ++k) {
      const dqmc_op& op = ops[k];
      const int32_t* i = op.i;
      bool ok = true;
      switch (op.kind) {
        case DQMC_OP_FEAT_EN: ok = okb(i[0]) && bufs[i[0]].rows == N && bufs[i[0]].width >= 4 * sys.n_nuc + (i[2] ? 1 : 0); break;
        case DQMC_OP_FEAT_EE: ok = okb(i[0]) && bufs[i[0]].rows == i[2] && bufs[i[0]].width == 4 && i[1] >= 0 && (size_t)(i[1] + 2 * i[2]) <= n_itable; break;
        case DQMC_OP_LINEAR: {
          ok = i[0] >= 1 && i[0] <= 4 && okb(i[17]);
          size_t wrows = 0;
          for (int p = 0; ok && p < i[0]; ++p) {
            const int sb = i[1 + 4 * p], r0 = i[2 + 4 * p], K = i[3 + 4 * p], bc = i[4 + 4 * p];
            ok = okb(sb) && K >= 1 && pad4(K) <= bufs[sb].width && r0 >= 0 && r0 + (bc ? 1 : i[20]) <= bufs[sb].rows;
            wrows += pad4(K);
          }
          if (ok) {
            const dqmc_buf& d = bufs[i[17]];
            const int ldw = pad4(i[21]);
            ok = i[18] >= 0 && i[18] + i[20] <= d.rows && i[19] >= 0 && i[19] % 4 == 0 && i[19] + ldw <= d.width &&
                 i[22] >= 0 && i[22] % 4 == 0 && (size_t)i[22] + wrows * ldw <= n_weights &&
                 (i[23] < 0 || (i[23] % 4 == 0 && (size_t)(i[23] + ldw) <= n_weights)) && i[24] >= 0 && i[24] <= 4;
            if (ok && i[25] >= 0)
              ok = okb(i[25]) && bufs[i[25]].width >= i[19] + ldw && i[26] >= 0 && i[26] + i[20] <= bufs[i[25]].rows;
          }
          break;
        }
        case DQMC_OP_SPIN_MEAN: ok = okb(i[0]) && okb(i[1]) && bufs[i[0]].rows == N && bufs[i[1]].rows == 2 && bufs[i[0]].width == bufs[i[1]].width; break;
        case DQMC_OP_CONST: ok = okb(i[0]) && i[1] >= 0 && (size_t)i[1] + (size_t)bufs[i[0]].rows * bufs[i[0]].width <= n_weights; break;
        case DQMC_OP_CONV: ok = okb(i[0]) && okb(i[1]) && okb(i[2]) && bufs[i[2]].rows == N && i[6] <= bufs[i[0]].width && i[6] <= bufs[i[1]].width && i[3] + i[6] <= bufs[i[2]].width && (size_t)(i[4] + 2 * N * i[5]) <= n_itable; break;
        case DQMC_OP_EDGE_SUM: ok = okb(i[0]) && okb(i[2]) && bufs[i[2]].rows == N && i[6] <= bufs[i[0]].width && i[3] + i[6] <= bufs[i[2]].width && (size_t)(i[4] + 2 * N * i[5]) <= n_itable; break;
        case DQMC_OP_ROW_SUM: ok = okb(i[0]) && okb(i[1]) && bufs[i[1]].rows == 1 && bufs[i[0]].width == bufs[i[1]].width; break;
        case DQMC_OP_ORBITALS: {
          if (i[6] < 0) { ok = false; break; }
          const size_t ne = (size_t)sys.n_det * N * sys.n_nuc * (i[6] > 0 ? i[6] : 1);
          ok = okb(i[0]) && okb(i[1]) && bufs[i[0]].rows == N && bufs[i[0]].width >= sys.n_det * N && bufs[i[1]].rows == sys.n_det && bufs[i[1]].width >= N * N;
          for (int q = 2; ok && q < 6; ++q) ok = i[q] >= 0 && (size_t)i[q] + ne <= n_weights;
          break;
        }
        case DQMC_OP_SLOGDET: ok = okb(i[0]) && bufs[i[0]].rows == sys.n_det && bufs[i[0]].width >= N * N && N <= 44; break;
        case DQMC_OP_FINAL: ok = (i[0] < 0 || (okb(i[0]) && bufs[i[0]].rows == 1)) && (i[1] < 0 || (size_t)(i[1] + sys.n_det) <= n_weights) && i[3] >= 0 && (size_t)(i[3] + 2) <= n_weights && i[2] >= 0 && i[2] <= 2; break;
        case DQMC_OP_ATTENTION:
          ok = i[4] >= 1 && i[5] >= 1;
          for (int q = 0; ok && q < 4; ++q) ok = okb(i[q]) && bufs[i[q]].rows == N && bufs[i[q]].width >= i[4] * i[5];
          if (ok) ok = bufs[i[0]].width == bufs[i[1]].width && bufs[i[0]].width == bufs[i[2]].width && bufs[i[0]].width == bufs[i[3]].width;
          if (ok) ok = i[6] >= 0 && (i[6] == 0 || (i[7] >= 0 && i[8] >= 0 && (size_t)i[7] + (size_t)i[6] * i[4] * i[5] <= n_weights &&
                                                   (size_t)i[8] + (size_t)i[6] * i[4] * i[5] <= n_weights));
          if (ok && dqmc::attention_lds_bytes<real>(N, i[5], i[6]) > (size_t)160 * 1024)
            return fail(DQMC_E_UNSUPPORTED, "attention tile set (N, head_dim) exceeds the 160 KiB LDS");
          break;
        default: ok = false;
      }
      if (ok && op.kind == DQMC_OP_FEAT_EE)          // senders: electron s >= 0, or nucleus -1 - s
        for (int r = 0; r < i[2] && ok; ++r) {
          const int rc = h_itable[i[1] + 2 * r], sd = h_itable[i[1] + 2 * r + 1];
          ok = rc >= 0 && rc < N && sd < N && -1 - sd < sys.n_nuc;
        }
      if (ok && op.kind == DQMC_OP_CONV)
        for (int q = 0; q < N * i[5] && ok; ++q) {
          const int row = h_itable[i[4] + 2 * q], sd = h_itable[i[4] + 2 * q + 1];
          if (row < 0) continue;
          ok = row < bufs[i[0]].rows && (sd >= 0 ? sd : -1 - sd) < bufs[i[1]].rows;
        }
      if (!ok) return fail(DQMC_E_ARG, "malformed op #" + std::to_string(k) + " kind " + std::to_string(op.kind));
    }
    return DQMC_OK;
  }

  // Row-wise two-layer MLPs (hkext.MLP with one hidden layer: the edge MLPs w / u and the node MLP h of a message-passing
  // layer): LINEAR op k writes a private hidden buffer that exactly one later LINEAR op reads as its single piece, same
  // rows, whole width.  The pair runs as one launch of the chained kernel at k's position, which is legal when nothing
  // between the two ops touches the second layer's output and its residual input is complete before k.
  void analyse_chains() {
    const int no = (int)ops.size(), nb = (int)bufs.size();
    mlp_child.assign(no, -1);
    mlp_skip.assign(no, 0);
    if (!mlp_fuse) return;
    std::vector<int> rd, wr;
    std::vector<std::vector<int>> writers(nb), readers(nb);
    for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) writers[b].push_back(k); for (int b : rd) readers[b].push_back(k); }
    for (int c = 0; c < no; ++c) {
      const int32_t* ci = ops[c].i;
      if (ops[c].kind != DQMC_OP_LINEAR || ci[0] != 1 || ci[4]) continue;                 // one non-broadcast piece
      const int hb = ci[1];
      if (writers[hb].size() != 1 || readers[hb].size() != 1) continue;                   // a private hidden buffer
      const int p = writers[hb][0];
      if (p >= c || ops[p].kind != DQMC_OP_LINEAR || mlp_child[p] >= 0 || mlp_skip[p]) continue;
      const int32_t* pi = ops[p].i;
      bool bc = false;
      for (int q = 0; q < pi[0]; ++q) bc = bc || pi[4 + 4 * q];
      if (bc || pi[25] >= 0) continue;                                                    // no broadcast pieces, no residual on the hidden layer
      if (pi[18] != ci[2] || pi[20] != ci[20] || pi[19] != 0 || pad4(ci[3]) != pad4(pi[21])) continue;   // same rows, whole width
      if (pad4(pi[21]) > 64 || pad4(ci[21]) > 32) continue;
      if ((pi[24] & 7) > 2 || (ci[24] & 7) > 4) continue;
      bool ok = true;
      if (ci[25] >= 0) for (int w : writers[ci[25]]) ok = ok && w < p;                    // residual input complete before the pair runs
      for (int m = p + 1; m < c && ok; ++m) {                                             // nobody in between reads or writes the output buffer
        op_io(ops[m], rd, wr);
        for (int b : rd) ok = ok && b != ci[17];
        for (int b : wr) ok = ok && b != ci[17];
      }
      if (!ok) continue;
      mlp_child[p] = c;
      mlp_skip[c] = 1;
    }
  }

  // The float64 tail (engine.hip, above tail_f64): the LINEAR ops that write the buffer ORBITALS reads (the backflow head),
  // ORBITALS, SLOGDET, FINAL.  None if the program has no ORBITALS op, if a writer of that buffer is not a plain LINEAR op
  // without residual (a chained MLP's hidden tile never reaches memory), or if a head op reads what a tail op writes.
  void analyse_tail() {
    const int no = (int)ops.size(), nb = (int)bufs.size();
    k_tail = -1; tail_in.clear(); in_tail.assign(no, 0);
    tail_written.assign(nb, 0); tail_direct.assign(nb, 0); tail_alloc.assign(nb, 0);
    int bf = -1;
    for (int k = 0; k < no; ++k) if (ops[k].kind == DQMC_OP_ORBITALS) bf = ops[k].i[0];
    if (bf < 0) return;
    std::vector<int> rd, wr;
    int first = -1;
    for (int k = 0; k < no; ++k) {
      op_io(ops[k], rd, wr);
      bool w_bf = false;
      for (int b : wr) w_bf = w_bf || b == bf;
      if (w_bf) {
        if (ops[k].kind != DQMC_OP_LINEAR || ops[k].i[25] >= 0) return;
        if (mlp_skip.size() == (size_t)no && (mlp_skip[k] || mlp_child[k] >= 0)) return;
        in_tail[k] = 1;
        if (first < 0) first = k;
      } else if (ops[k].kind == DQMC_OP_ORBITALS || ops[k].kind == DQMC_OP_SLOGDET || ops[k].kind == DQMC_OP_FINAL) {
        in_tail[k] = 1;
      }
    }
    if (first <= 0) { in_tail.assign(no, 0); return; }
    std::vector<char> head_w(nb, 0), tail_r(nb, 0), nonlin_r(nb, 0);
    for (int k = 0; k < no; ++k) {
      op_io(ops[k], rd, wr);
      if (in_tail[k]) {
        for (int b : wr) tail_written[b] = 1;
        for (int b : rd) { tail_r[b] = 1; if (ops[k].kind != DQMC_OP_LINEAR) nonlin_r[b] = 1; }
      } else {
        for (int b : wr) head_w[b] = 1;
      }
    }
    for (int k = 0; k < no; ++k) {               // the head must not depend on the tail; tail ops must not write head buffers
      if (in_tail[k]) continue;
      op_io(ops[k], rd, wr);
      for (int b : rd) if (tail_written[b]) { in_tail.assign(no, 0); return; }
    }
    for (int b = 0; b < nb; ++b) {
      if (tail_written[b] && head_w[b]) { in_tail.assign(no, 0); tail_written.assign(nb, 0); return; }
      if (tail_r[b] && head_w[b]) { tail_in.push_back(b); tail_direct[b] = !nonlin_r[b]; }
    }
    // a tail LINEAR op reads its pieces either all as float32 (direct) or all from the twin's workspace
    // (to a fixpoint: demoting a buffer for one op can leave an EARLIER op that shares it mixed)
    for (bool changed = true; changed;) {
      changed = false;
      for (int k = 0; k < no; ++k) {
        if (!in_tail[k] || ops[k].kind != DQMC_OP_LINEAR) continue;
        bool all_direct = true, any_direct = false;
        for (int p = 0; p < ops[k].i[0]; ++p) { const bool d = tail_direct[ops[k].i[1 + 4 * p]]; all_direct = all_direct && d; any_direct = any_direct || d; }
        if (!all_direct && any_direct) {
          for (int p = 0; p < ops[k].i[0]; ++p) tail_direct[ops[k].i[1 + 4 * p]] = 0;
          changed = true;
        }
      }
    }
    for (int b = 0; b < nb; ++b) tail_alloc[b] = tail_written[b] || (tail_r[b] && head_w[b] && !tail_direct[b]);
    k_tail = first;
  }

  // Stream slots of the Laplacian pass.  Edge-stream ops (pair-compact destination) keep slot 1.  Every other op goes, in
  // program order, to a slot whose last op it depends on anyway (directly or through other ops) -- placing it there costs
  // no concurrency -- preferring the main slot, then the slot of its most recent producer, then a free one; only if
  // there is none does it queue behind unrelated work on the main slot.  A chained MLP pair counts as one op.
  void analyse_streams() {
    const int no = (int)ops.size(), nb = (int)bufs.size();
    op_sid.assign(no, 0);
    std::vector<int> rd, wr, rd2, wr2;
    std::vector<std::vector<int>> writers(nb);
    std::vector<std::vector<char>> dep(no, std::vector<char>(no, 0));
    auto io = [&](int k) {
      op_io(ops[k], rd, wr);
      if (mlp_child.size() == (size_t)no && mlp_child[k] >= 0) {
        op_io(ops[mlp_child[k]], rd2, wr2);
        for (int b : rd2) if (b != ops[k].i[17]) rd.push_back(b);
        for (int b : wr2) wr.push_back(b);
      }
    };
    int last[4] = {-1, -1, -1, -1};
    for (int k = 0; k < no; ++k) {
      if (mlp_skip.size() == (size_t)no && mlp_skip[k]) {     // rides with its parent (same slot if it has to run on its own)
        op_sid[k] = 0;
        for (int p = 0; p < k; ++p) if (mlp_child[p] == k) op_sid[k] = op_sid[p];
        continue;
      }
      io(k);
      int producer = -1;
      for (int b : rd)
        for (int w : writers[b]) {
          if (w >= k) continue;
          dep[k][w] = 1;
          for (int x = 0; x < w; ++x) if (dep[w][x]) dep[k][x] = 1;
          if (w > producer) producer = w;
        }
      const dqmc_op& o = ops[k];
      const bool edge = (o.kind == DQMC_OP_FEAT_EE && compact[o.i[0]]) || (o.kind == DQMC_OP_LINEAR && compact[o.i[17]]);
      int sid = 0;
      if (edge) sid = 1;
      else if (o.kind == DQMC_OP_SLOGDET || o.kind == DQMC_OP_FINAL || o.kind == DQMC_OP_ATTENTION) sid = 0;
      else {
        auto eligible = [&](int s_) { return last[s_] < 0 || dep[k][last[s_]]; };
        if (eligible(0)) sid = 0;
        else if (producer >= 0 && op_sid[producer] >= 2 && eligible(op_sid[producer])) sid = op_sid[producer];
        else if (eligible(2)) sid = 2;
        else if (eligible(3)) sid = 3;
        else sid = 0;
      }
      op_sid[k] = sid;
      last[sid] = k;
      for (int b : wr) writers[b].push_back(k);
    }
  }

  // Which buffers can carry pair-compact lanes: outputs of FEAT_EE and of row-wise LINEAR ops on them (all
  // pieces compact, none broadcast; residual compact with the same row pairs).  Consumers that understand the
  // compact layout: LINEAR, CONV (edge operand), EDGE_SUM.  Anything else turns the optimisation off.
  void analyse_lanes() {
    const int nb = (int)bufs.size();
    compact.assign(nb, 0);
    pair_rs.assign(nb, std::vector<int>());
    if (!lane_compact) return;
    std::vector<char> full_written(nb, 0);
    bool ok = true;
    auto set_pairs = [&](int b, int row, int rc, int sd) {
      if (pair_rs[b].empty()) pair_rs[b].assign(2 * (size_t)bufs[b].rows, -2);
      int& pr = pair_rs[b][2 * row];
      int& ps = pair_rs[b][2 * row + 1];
      if (pr != -2 && (pr != rc || ps != sd)) ok = false;
      pr = rc; ps = sd;
    };
    for (size_t k = 0; k < ops.size() && ok; ++k) {
      const int32_t* i = ops[k].i;
      switch (ops[k].kind) {
        case DQMC_OP_FEAT_EE:
          if (full_written[i[0]]) { ok = false; break; }
          compact[i[0]] = 1;
          for (int r = 0; r < i[2]; ++r) set_pairs(i[0], r, h_itable[i[1] + 2 * r], h_itable[i[1] + 2 * r + 1]);
          break;
        case DQMC_OP_LINEAR: {
          int n_c = 0;
          for (int p = 0; p < i[0]; ++p) n_c += compact[i[1 + 4 * p]] ? 1 : 0;
          if (n_c == 0) {
            if (compact[i[17]] || (i[25] >= 0 && compact[i[25]])) ok = false;
            full_written[i[17]] = 1;
            break;
          }
          if (n_c != i[0] || full_written[i[17]]) { ok = false; break; }
          compact[i[17]] = 1;
          for (int rr = 0; rr < i[20] && ok; ++rr) {
            const int sb0 = i[1], r00 = i[2];
            if (i[4] || pair_rs[sb0].empty()) { ok = false; break; }
            const int rc = pair_rs[sb0][2 * (r00 + rr)], sd = pair_rs[sb0][2 * (r00 + rr) + 1];
            for (int p = 1; p < i[0]; ++p) {
              const int sb = i[1 + 4 * p], r0 = i[2 + 4 * p];
              if (i[4 + 4 * p] || pair_rs[sb].empty() || pair_rs[sb][2 * (r0 + rr)] != rc || pair_rs[sb][2 * (r0 + rr) + 1] != sd) ok = false;
            }
            if (i[25] >= 0) {
              const int rb = i[25];
              if (!compact[rb] || pair_rs[rb].empty() || pair_rs[rb][2 * (i[26] + rr)] != rc || pair_rs[rb][2 * (i[26] + rr) + 1] != sd) ok = false;
            }
            set_pairs(i[17], i[18] + rr, rc, sd);
          }
          break;
        }
        case DQMC_OP_CONV:
          if (compact[i[1]] || compact[i[2]]) ok = false;
          full_written[i[2]] = 1;
          if (compact[i[0]])      // the table's (row, sender) of receiver el must be the row's own pair
            for (int el = 0; el < N && ok; ++el)
              for (int sdx = 0; sdx < i[5]; ++sdx) {
                const int row = h_itable[i[4] + 2 * (el * i[5] + sdx)], snd = h_itable[i[4] + 2 * (el * i[5] + sdx) + 1];
                if (row >= 0 && (pair_rs[i[0]][2 * row] != el || pair_rs[i[0]][2 * row + 1] != snd)) ok = false;
              }
          break;
        case DQMC_OP_EDGE_SUM:
          if (compact[i[2]]) ok = false;
          full_written[i[2]] = 1;
          if (compact[i[0]])
            for (int el = 0; el < N && ok; ++el)
              for (int sdx = 0; sdx < i[5]; ++sdx) {
                const int row = h_itable[i[4] + 2 * (el * i[5] + sdx)], snd = h_itable[i[4] + 2 * (el * i[5] + sdx) + 1];
                if (row >= 0 && (pair_rs[i[0]][2 * row] != el || pair_rs[i[0]][2 * row + 1] != snd)) ok = false;
              }
          break;
        case DQMC_OP_FEAT_EN: case DQMC_OP_CONST: full_written[i[0]] = 1; break;
        case DQMC_OP_SPIN_MEAN: case DQMC_OP_ROW_SUM: if (compact[i[0]]) ok = false; full_written[i[1]] = 1; break;
        case DQMC_OP_ORBITALS: if (compact[i[0]]) ok = false; full_written[i[1]] = 1; break;
        case DQMC_OP_SLOGDET: if (compact[i[0]]) ok = false; break;
        case DQMC_OP_FINAL: if (i[0] >= 0 && compact[i[0]]) ok = false; break;
        case DQMC_OP_ATTENTION:
          for (int q = 0; q < 3; ++q) if (compact[i[q]]) ok = false;
          full_written[i[3]] = 1;
          break;
        default: break;
      }
    }
    for (int b = 0; b < nb && ok; ++b)
      if (compact[b]) for (int v : pair_rs[b]) if (v == -2) ok = false;   // every row of a compact buffer has a pair
    if (!ok) { compact.assign(nb, 0); pair_rs.assign(nb, std::vector<int>()); }
  }
  // lanes of buffer b in an evaluation with TP lanes
  int lanes_of(int b, int TP) const { return (TP > 1 && compact[b]) ? dqmc::PAIR_LANES : TP; }

  int set_weights(const double* w, size_t n) override {
    if (n != n_weights) return fail(DQMC_E_ARG, "weight buffer length differs from the one given at creation");
    ++graph_epoch;
    wtmp.resize(n);
    for (size_t k = 0; k < n; ++k) wtmp[k] = (real)w[k];
    if (sizeof(real) == 4) {
      w64_h.assign(w, w + n);
      if (twin) { const int rc = twin->set_weights(w, n); if (rc) return rc; }
    }
    HIP_TRY(hipMemcpyAsync(d_w, wtmp.data(), sizeof(real) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int rc_t = pack_spec_tape();
    if (rc_t) return rc_t;
    if (fused_n_ops > 0) return pack_fused_weights();
    return DQMC_OK;
  }
  // FNV-1a over the integers that define the program's structure (deepqmc_amd/codegen/substep.py: program_hash computes the
  // same value when it generates a plan-specialised kernel)
  uint64_t program_hash() const {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](int32_t v) {
      for (int k = 0; k < 4; ++k) { h = (h ^ (uint64_t)((uint32_t)v >> (8 * k) & 0xffu)) * 0x100000001b3ull; }
    };
    mix(sys.n_up); mix(sys.n_down); mix(sys.n_nuc); mix(sys.n_det); mix((int32_t)bufs.size()); mix((int32_t)ops.size()); mix((int32_t)n_itable);
    for (const auto& b : bufs) { mix(b.rows); mix(b.width); }
    for (const auto& o : ops) {
      mix(o.kind);
      for (int k = 0; k < 28; ++k) mix(o.i[k]);
      for (int k = 0; k < 4; ++k) { int32_t bits; memcpy(&bits, &o.f[k], 4); mix(bits); }
    }
    for (int32_t v : h_itable) mix(v);
    return h;
  }
  int pack_spec_tape() {
    if constexpr (sizeof(real) == 4) {
      if (!spec_k) return DQMC_OK;
      std::vector<uint32_t> tape((size_t)spec_k->tape_bytes / 4);
      dqmc::spec_pack_tape(*spec_k, reinterpret_cast<const float*>(wtmp.data()), tape.data());
      if (!d_tape) HIP_TRY(hipMalloc((void**)&d_tape, (size_t)spec_k->tape_bytes));
      HIP_TRY(hipMemcpyAsync(d_tape, tape.data(), (size_t)spec_k->tape_bytes, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    return DQMC_OK;
  }
  // ---- fused value-only evaluation (kernel_fused2.hip) ----------------------------------
  // Ops [0, fused_n_ops) (everything up to and including ORBITALS) run in one kernel on a tile
  // of WT walkers with LDS-resident buffers; buffers read by later ops stay in the workspace.
  int option(const char* name, int value) override {
    const std::string s(name);
    if (s.rfind("twin.", 0) == 0) {            // an option of the float64 refinement twin (applied when it is created, too)
      twin_opts.emplace_back(s.substr(5), value);
      return twin ? twin->option(s.c_str() + 5, value) : DQMC_OK;
    }
    ++graph_epoch;                             // (any switch may change what a captured pass would launch)
    if (s == "pass_graph") { pass_graph = value; if (!value) drop_graphs(); if (twin) twin->option("pass_graph", value); twin_opts.emplace_back(s, value); return DQMC_OK; }
    if (s == "fused") { fused_enabled = value; return DQMC_OK; }
    if (s == "fused_wt") { fused_wt_req = value; return build_fused_plan(); }
    if (s == "fused_occ") { fused_occ_req = value; return DQMC_OK; }
    if (s == "attention_mfma") { attention_mfma = value; return DQMC_OK; }
    if (s == "attention_ncb") { attention_ncb = value; return DQMC_OK; }
    if (s == "attention_split") { attention_split = value; return DQMC_OK; }
    if (s == "ws_budget_mb") { if (value < 1) return fail(DQMC_E_ARG, "ws_budget_mb must be positive"); ws_budget = (size_t)value << 20; return DQMC_OK; }
    if (s == "slogdet_mfma") { slogdet_mfma = value; return DQMC_OK; }
    if (s == "lane_compact") { lane_compact = value != 0; analyse_lanes(); last_B = 0; return DQMC_OK; }
    if (s == "linear_f64_split") { linear_f64_split = value; return DQMC_OK; }
    if (s == "linear_bkx") { linear_bkx = value; return DQMC_OK; }      // (kernel selection of kernel_linear.hip, this context only)
    if (s == "mlp_fuse") { mlp_fuse = value; analyse_chains(); analyse_tail(); return DQMC_OK; }
    if (s == "tail_f64") { tail_f64 = value; return DQMC_OK; }
    if (s == "no_twin") { if (value && twin) return fail(DQMC_E_ARG, "no_twin must be set before the first local-energy call creates the twin"); no_twin = value; return DQMC_OK; }
    if (s == "fused_substep") { fused_substep = value; return DQMC_OK; }
    if (s == "fused_spec") { fused_spec = value; return DQMC_OK; }
    if (s == "split_bcast") { split_bcast = value; return DQMC_OK; }
    if (s == "dual_stream") { dual_stream = value; return DQMC_OK; }
    if (s == "multi_stream") { multi_stream = value; return DQMC_OK; }
    if (s == "fused_prio") { fused_prio = value; return DQMC_OK; }
    if (s == "fused_lean") { fused_lean = value; return build_fused_plan(); }
    if (s == "fused_bf") { fused_bf = value; return build_fused_plan(); }
    if (s == "linear_bf") { linear_bf = value; return DQMC_OK; }      // (kernel selection of kernel_linear.hip, this context only)
    if (s == "fused_wg_per_cu") {
      if (value < 4 || value > 6) return fail(DQMC_E_ARG, "fused_wg_per_cu must be 4, 5 or 6");
      fused2_lds_quarter = (size_t)160 * 1024 / value;
      return build_fused_plan();
    }
    if (s == "refine") { refine = value; return DQMC_OK; }
    if (s == "twin_full_budget") { twin_full_budget = value; if (twin) twin->option("ws_budget_mb", (int)((value ? ws_budget : ws_budget / 2) >> 20)); return DQMC_OK; }
    if (s == "refine_sample") { if (value < 2) return fail(DQMC_E_ARG, "refine_sample must be >= 2"); refine_sample = value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_probe") { if (value < 0) return fail(DQMC_E_ARG, "refine_probe must be >= 0"); refine_probe = value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_miss_e9") { if (value < 1) return fail(DQMC_E_ARG, "refine_miss_e9 must be >= 1"); refine_miss = 1e-9 * value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_target_e7") { if (value < 1) return fail(DQMC_E_ARG, "refine_target_e7 must be >= 1"); refine_target = 1e-7 * value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_direct_calls") { if (value < 0) return fail(DQMC_E_ARG, "refine_direct_calls must be >= 0"); refine_direct_calls = value; return DQMC_OK; }
    if (s == "refine_direct_backoff") { if (value < 0 || value > 16) return fail(DQMC_E_ARG, "refine_direct_backoff must be in [0, 16]"); refine_direct_backoff = value; return DQMC_OK; }
    if (s == "refine_direct_pct") { if (value < 1 || value > 100) return fail(DQMC_E_ARG, "refine_direct_pct must be 1..100"); refine_direct_enter = 0.01 * value; if (refine_direct_exit > refine_direct_enter) refine_direct_exit = refine_direct_enter; return DQMC_OK; }
    if (s == "refine_direct_exit_pct") { if (value < 0 || value > 100) return fail(DQMC_E_ARG, "refine_direct_exit_pct must be 0..100"); refine_direct_exit = 0.01 * value; if (refine_direct_exit > refine_direct_enter) refine_direct_enter = refine_direct_exit; return DQMC_OK; }
    if (s == "refine_thresh") { if (value < 0) return fail(DQMC_E_ARG, "refine_thresh must be >= 0"); refine_thresh = (double)value; return DQMC_OK; }
    if (s == "fused_sched_kb") { fused_sched_budget = (size_t)value * 1024; return build_fused_plan(); }
    if (s == "fused_print") {   // plan summary on stderr (tuning aid)
      fprintf(stderr, "[dqmc] fused plan: WT=%d lds=%zu B; %d fused ops, %d levels\n", fused2_WT, fused2_lds, fused_n_ops,
              fused_n_ops ? f_level[fused_n_ops - 1] + 1 : 0);
      if (value >= 2)        // one line per descriptor of every wave list: kind, scheduled op, its op kind, row blocks, quads of k-steps
        for (size_t w = 0; w < plan_lists.size(); ++w)
          for (size_t k = 0; k < plan_lists[w].size(); ++k) {
            const dqmc::FDesc& d = plan_lists[w][k];
            int nq = 0;
            for (int p = 0; p < d.n_pieces; ++p) nq += d.a_nq[p];
            fprintf(stderr, "[dqmc] wave %zu desc %zu kind %d op %d opkind %d ma %d row0 %d col0 %d quads %d ldw %d\n", w, k, d.kind, d.op,
                    d.kind == 2 ? 0 : ops[f_order[d.op]].kind, d.ma, d.row0, d.col0, nq, d.ldw);
          }
      return DQMC_OK;
    }
    if (s == "ecp_mixed") { ecp_mixed_on = value; return DQMC_OK; }
    if (s == "ecp_heavy_e6") { if (value < 0) return fail(DQMC_E_ARG, "ecp_heavy_e6 must be >= 0"); ecp_w_heavy = 1e-6 * value; return DQMC_OK; }
    if (s == "ecp_dlog_floor_e6") { if (value < 0) return fail(DQMC_E_ARG, "ecp_dlog_floor_e6 must be >= 0"); ecp_dlog_floor = 1e-6 * value; return DQMC_OK; }
    if (s == "ecp_skip_e12") { if (value < 0) return fail(DQMC_E_ARG, "ecp_skip_e12 must be >= 0"); ecp_w_skip = 1e-12 * value; return DQMC_OK; }
    if (s == "ecp_max_cfg") { if (value < 1) return fail(DQMC_E_ARG, "ecp_max_cfg must be positive"); ecp_max_cfg = (size_t)value; return DQMC_OK; }
    if (s == "fused_sched") { fused_sched_mode = value; return build_fused_plan(); }
    if (s == "fused_dbg") {
      fused_dbg = value;
      if (value && !d_prof) { HIP_TRY(hipMalloc((void**)&d_prof, sizeof(long long) * (9 * ops.size() + 80 + 1024 + 2 * 8192))); HIP_TRY(hipMemsetAsync(d_prof, 0, sizeof(long long) * (9 * ops.size() + 80 + 1024 + 2 * 8192), st)); }
      return DQMC_OK;
    }
    if (s == "fused_lds_kb") { fused2_lds_budget = (size_t)value * 1024; return build_fused_plan(); }
    return fail(DQMC_E_ARG, "unknown option " + s);
  }

  // LDS placement of the buffers for a tile of WT walkers: first-fit over live intervals.
  // Buffers read / written by an op (buffer granularity; partial writers of one buffer touch
  // disjoint row or column ranges by construction of the program).
