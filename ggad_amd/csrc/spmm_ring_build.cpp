// Host half of the LDS-RING product (k_spmm_ring in fullgraph.hip): the schedule of its entry stream, on threads.
//
// The operand slice (n_src rows x 32 floats) passes through a ring of S slots of slot_rows rows in LDS: during PHASE j the slots
// j .. j+V-1 (the WINDOW) are resident and a loader wave fetches slot j+S-1 into the buffer that phase j-1 released, so staging
// overlaps the walk and there is one barrier per phase instead of two per panel.  A round is 8 lane rows (lane group g of a wave
// accumulates lane row g: a matrix row, or -- WIDE rounds, for the hub rows -- an eighth of one row's entries, dealt by the parity
// of their LDS row to the groups {0,1,4,5} / {3,2,7,6}).  The schedule is FLEXIBLE: in phase j a round walks
// T = 4 * ceil(max_g(mandatory_g) / 4) steps, where mandatory_g = entries of lane row g in slot j that are still open (slot j is
// overwritten after the phase); a group with fewer open entries in slot j fills its steps with its next entries anywhere in the
// window (from the LDS-row parity it has fewer of first: bank-sharing partners alternate parities), which lowers its mandatory count of
// the following phases.  The groups of a round therefore drift apart by up to a
// window instead of being padded to the longest row of every panel: 0.73 -> 0.85 of the step slots carry an entry on the
// T-Finance-size power-law graph (scripts/ring_fill_sim.py).  What the kernel sees is a flat per-wave list of QUADS (4 steps x 8
// lane groups x 16-bit LDS row index), each tagged with the accumulator it adds to and an end-of-phase flag.
//
// ggad_spmm_ring_count: quads of every (round, phase).  ggad_spmm_ring_fill: the indices, given where each tile starts.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int GROUPS = 8;
// lane groups whose 128-byte rows share a 16-lane service group of ds_read_b128: (0,3) (1,2) (4,7) (5,6)
constexpr int PARTNER[GROUPS] = {3, 2, 1, 0, 7, 6, 5, 4};
constexpr bool FIRST_ODD[GROUPS] = {false, false, true, true, false, false, true, true};
constexpr int EVEN_GROUPS[4] = {0, 1, 4, 5}, ODD_GROUPS[4] = {3, 2, 7, 6};

template <class F>
void parallel_rounds(int32_t n_rounds, int32_t n_threads, F &&body) {
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, 32));
  nt = std::min<int>(nt, std::max(1, n_rounds / 64));
  if (nt == 1) { body(0, n_rounds); return; }
  std::vector<std::thread> th;
  const int32_t per = (n_rounds + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int32_t a = t * per, b = std::min(n_rounds, a + per);
    if (a >= b) break;
    th.emplace_back([=, &body] { body(a, b); });
  }
  for (auto &x : th) x.join();
}

struct Ring {
  int32_t slot_rows, S, V, n_phases;
  int32_t lds_row(int32_t c) const { return ((c / slot_rows) % S) * slot_rows + c % slot_rows; }
};

// the 8 lane rows of a round as sorted column lists, one per parity of the LDS row an entry is read from
struct LaneRows {
  std::vector<int32_t> e[GROUPS][2];
  void load(const int64_t *rowptr, const int32_t *col, const int32_t *rows, bool wide, bool skip_diag, const Ring &R) {
    for (int g = 0; g < GROUPS; ++g) { e[g][0].clear(); e[g][1].clear(); }
    if (wide) {
      const int32_t row = rows[0];
      if (row < 0) return;
      int n_even = 0, n_odd = 0;
      for (int64_t i = rowptr[row]; i < rowptr[row + 1]; ++i) {
        const int32_t c = col[i];
        if (skip_diag && c == row) continue;
        if (R.lds_row(c) & 1) e[ODD_GROUPS[n_odd++ & 3]][1].push_back(c);
        else e[EVEN_GROUPS[n_even++ & 3]][0].push_back(c);
      }
      return;
    }
    for (int g = 0; g < GROUPS; ++g) {
      const int32_t row = rows[g];
      if (row < 0) continue;
      for (int64_t i = rowptr[row]; i < rowptr[row + 1]; ++i)
        if (!(skip_diag && col[i] == row)) e[g][R.lds_row(col[i]) & 1].push_back(col[i]);
    }
  }
};

// walks the phases of one round; tile(j, T, pos[g][par], take[g][par]) is called for every phase with T > 0.  A group takes all its
// open entries of slot j (both parities) and fills the T steps up from the rest of the window -- from the parity it has fewer of first,
// so that it can alternate parities against its bank-sharing partner for as long as possible.
template <class Tile>
void schedule_round(const LaneRows &L, const Ring &R, Tile &&tile) {
  size_t pos[GROUPS][2] = {}, mend[GROUPS][2] = {}, wend[GROUPS][2] = {};
  int32_t take[GROUPS][2];
  for (int32_t j = 0; j < R.n_phases; ++j) {
    const int64_t m_hi = (int64_t)(j + 1) * R.slot_rows, w_hi = (int64_t)(j + R.V) * R.slot_rows;
    int32_t need = 0;
    for (int g = 0; g < GROUPS; ++g) {
      int32_t m = 0;
      for (int par = 0; par < 2; ++par) {
        const auto &v = L.e[g][par];
        size_t &me = mend[g][par], &we = wend[g][par];
        while (me < v.size() && v[me] < m_hi) ++me;
        if (we < me) we = me;
        while (we < v.size() && v[we] < w_hi) ++we;
        if (me > pos[g][par]) m += (int32_t)(me - pos[g][par]);      // (pos may be ahead of the slot: earlier fill-ins)
      }
      need = std::max(need, m);
    }
    if (need == 0) continue;
    const int32_t T = (need + 3) / 4 * 4;
    for (int g = 0; g < GROUPS; ++g) {
      int32_t n[2] = {mend[g][0] > pos[g][0] ? (int32_t)(mend[g][0] - pos[g][0]) : 0, mend[g][1] > pos[g][1] ? (int32_t)(mend[g][1] - pos[g][1]) : 0};
      const int32_t avail[2] = {(int32_t)(wend[g][0] - pos[g][0]), (int32_t)(wend[g][1] - pos[g][1])};
      int32_t spare = T - n[0] - n[1];
      while (spare > 0) {
        int par = n[0] <= n[1] ? 0 : 1;
        if (n[par] >= avail[par]) par ^= 1;
        if (n[par] >= avail[par]) break;
        ++n[par];
        --spare;
      }
      take[g][0] = n[0];
      take[g][1] = n[1];
    }
    tile(j, T, pos, take);
    for (int g = 0; g < GROUPS; ++g) { pos[g][0] += (size_t)take[g][0]; pos[g][1] += (size_t)take[g][1]; }
  }
}

}  // namespace

extern "C" {

// quads_rp[round * n_phases + phase] = quads (4 steps) the round walks in that phase.
int ggad_spmm_ring_count(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                         const int32_t *round_wide, int32_t skip_diag, int32_t slot_rows, int32_t n_ring_slots, int32_t window,
                         int32_t n_phases, uint16_t *quads_rp, int32_t n_threads) {
  if (!rowptr || !col || !round_rows || !quads_rp || n_rounds < 0 || slot_rows < 2 || n_ring_slots < 2 || window < 1 ||
      window >= n_ring_slots || n_phases < 1)
    return -1;
  const Ring R{slot_rows, n_ring_slots, window, n_phases};
  int bad = 0;
  parallel_rounds(n_rounds, n_threads, [&](int32_t r0, int32_t r1) {
    LaneRows L;
    for (int32_t r = r0; r < r1; ++r) {
      uint16_t *dst = quads_rp + (int64_t)r * n_phases;
      std::fill(dst, dst + n_phases, (uint16_t)0);
      L.load(rowptr, col, round_rows + (int64_t)r * GROUPS, round_wide && round_wide[r], skip_diag != 0, R);
      schedule_round(L, R, [&](int32_t j, int32_t T, const size_t (*)[2], const int32_t (*)[2]) {
        if (T / 4 > 0xffff) __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
        dst[j] = (uint16_t)(T / 4);
      });
    }
  });
  return bad ? -2 : 0;
}

// idx: uint16 [n_sb * 128]; super-block = 4 quads: [half (quads 2h, 2h+1)][lane group][quad in half][step].  Every slot is first set to
// a zero row (row n_ring_slots * slot_rows of the parity class of the group, so that bank-sharing partners never collide on padding);
// tile (r, j) starts at quad quad_off[r * n_phases + j] (absolute) and covers quads_rp[r * n_phases + j] quads.
int ggad_spmm_ring_fill(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                        const int32_t *round_wide, int32_t skip_diag, int32_t slot_rows, int32_t n_ring_slots, int32_t window,
                        int32_t n_phases, const uint16_t *quads_rp, const int64_t *quad_off, uint16_t *idx, int64_t n_sb,
                        int32_t n_threads) {
  if (!rowptr || !col || !round_rows || !quads_rp || !quad_off || !idx || slot_rows < 2 || (slot_rows & 1) || n_ring_slots < 2 ||
      window < 1 || window >= n_ring_slots || (int64_t)slot_rows * n_ring_slots + 2 > 65535)
    return -1;
  const Ring R{slot_rows, n_ring_slots, window, n_phases};
  const uint16_t zero_even = (uint16_t)(slot_rows * n_ring_slots), zero_odd = (uint16_t)(zero_even + 1);    // slot_rows is even
  auto at = [](int64_t quad, int g, int t) -> int64_t {                  // slot of (quad, group, step 0..3) in idx
    return (quad >> 2) * 128 + ((quad >> 1) & 1) * 64 + g * 8 + (quad & 1) * 4 + t;
  };
  {                                                                      // default content: padding
    const int nt = std::max(1, std::min<int>(16, (int)(n_sb / 4096)));
    std::vector<std::thread> th;
    const int64_t per = (n_sb + nt - 1) / nt;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([=] {
        for (int64_t sb = t * per; sb < std::min(n_sb, (t + 1) * per); ++sb)
          for (int h = 0; h < 2; ++h)
            for (int g = 0; g < GROUPS; ++g)
              for (int s = 0; s < 8; ++s) idx[sb * 128 + h * 64 + g * 8 + s] = FIRST_ODD[g] ? zero_odd : zero_even;
      });
    for (auto &x : th) x.join();
  }
  int bad = 0;
  parallel_rounds(n_rounds, n_threads, [&](int32_t r0, int32_t r1) {
    LaneRows L;
    for (int32_t r = r0; r < r1; ++r) {
      const bool wide = round_wide && round_wide[r];
      L.load(rowptr, col, round_rows + (int64_t)r * GROUPS, wide, skip_diag != 0, R);
      schedule_round(L, R, [&](int32_t j, int32_t T, const size_t (*pos)[2], const int32_t (*take)[2]) {
        const int64_t q0 = quad_off[(int64_t)r * n_phases + j];
        if (quads_rp[(int64_t)r * n_phases + j] != T / 4 || q0 < 0 || q0 + T / 4 > n_sb * 4) { __atomic_store_n(&bad, 1, __ATOMIC_RELAXED); return; }
        // order inside a tile: the two lane rows of a bank-sharing pair (their 128-byte rows meet in two 16-lane service groups of the
        // ds_read_b128) must read LDS rows of OPPOSITE parity in the same step.  Per pair and step, greedily: the parity assignment
        // (even, odd) or (odd, even) under which more of the two rows have an entry left (ties: the one that takes from the longer
        // lists); a row without entry of its parity reads the zero row of that parity if it can still afford a padded step, else it
        // takes an entry of the other parity (a collision: + 2 LDS cycles for that half wave).
        for (int g = 0; g < GROUPS; ++g) {
          const int q = PARTNER[g];
          if (g > q) continue;
          int32_t left[2][2] = {{take[g][0], take[g][1]}, {take[q][0], take[q][1]}};     // [row of the pair][parity]
          size_t at_e[2][2] = {{pos[g][0], pos[g][1]}, {pos[q][0], pos[q][1]}};
          const int grp[2] = {g, q};
          for (int32_t t = 0; t < T; ++t) {
            const int32_t rem = T - t;
            int best = 0, best_score = -1;
            for (int asg = 0; asg < 2; ++asg) {                            // asg 0: row 0 reads even, row 1 odd; asg 1: the other way
              int score = 0;
              for (int r = 0; r < 2; ++r) {
                const int par = asg ^ r;
                const int32_t tot = left[r][0] + left[r][1];
                if (left[r][par] > 0) score += 4 + (left[r][par] >= left[r][par ^ 1] ? 1 : 0);
                else if (tot >= rem) score -= 8;                          // no slack: this row would have to collide
              }
              if (score > best_score) { best_score = score; best = asg; }
            }
            for (int r = 0; r < 2; ++r) {
              int par = best ^ r;
              const int32_t tot = left[r][0] + left[r][1];
              uint16_t v;
              if (left[r][par] == 0 && tot >= rem && tot > 0) par ^= 1;     // forced: an entry of the other parity
              if (left[r][par] > 0) {
                v = (uint16_t)R.lds_row(L.e[grp[r]][par][at_e[r][par]++]);
                --left[r][par];
              } else {
                v = par ? zero_odd : zero_even;
              }
              idx[at(q0 + (t >> 2), grp[r], t & 3)] = v;
            }
          }
          if (left[0][0] | left[0][1] | left[1][0] | left[1][1]) __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
        }
      });
    }
  });
  return bad ? -2 : 0;
}

}  // extern "C"
