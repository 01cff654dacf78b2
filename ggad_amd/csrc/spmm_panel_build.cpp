// Host half of the LDS-panel product (k_spmm_panel in fullgraph.hip): the two O(nnz) passes of its plan, on threads.
//
// ggad_amd/fullgraph.py::Csr.panel_plan sorts the rows into rounds of 8, calls ggad_spmm_panel_count (entries of every row
// per panel of ggad_spmm_panel_rows() columns -> the longest row of every (round, panel)), deals the rounds to workgroups / waves and lays the
// tiles out (numpy on n_rounds x n_panels values), then calls ggad_spmm_panel_fill, which writes the entry stream the
// kernel walks: per tile [oct][lane group][step] 16-bit panel row indices.  A round is 8 rows (lane group g walks row g) or, for
// the hub rows, ONE row WIDE: its entries of a panel are dealt over all 8 lane groups (even panel rows to groups 0 1 4 5, odd to
// 3 2 7 6: the bank-sharing pairs read opposite halves by construction) and the kernel adds the 8 accumulators in the epilogue --
// a 7,000-entry row is then 900 steps of work for one wave instead of 7,000, and the longest round no longer exceeds a wave's share.  Both passes walk the CSR once, a thread per
// block of rounds (a round's 8 rows and its tiles belong to one thread: no shared writes).  numpy did the same in 1.3 s
// at 21 M entries -- more than the 500 epochs of a T-Finance run save; this takes a few tens of milliseconds.
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace {

constexpr int GROUPS = 8;
// lane groups whose 128-byte rows share a 16-lane service group of ds_read_b128: (0,3) (1,2) (4,7) (5,6)
constexpr int PARTNER[GROUPS] = {3, 2, 1, 0, 7, 6, 5, 4};
constexpr bool FIRST_ODD[GROUPS] = {false, false, true, true, false, false, true, true};
constexpr int EVEN_GROUPS[4] = {0, 1, 4, 5}, ODD_GROUPS[4] = {3, 2, 7, 6};   // wide rounds: where even / odd panel rows go

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

}  // namespace

extern "C" {

// steps_rc[round * n_panels + panel] = entries of the longest of the round's rows in that panel.
// round_rows[round * 8 + g] = row of lane group g, or -1.  skip_diag: entries with col == row are not part of the stream.
int ggad_spmm_panel_count(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                          const int32_t *round_wide, int32_t skip_diag, int32_t panel_rows, int32_t n_panels, int32_t *steps_rc,
                          int32_t n_threads) {
  if (!rowptr || !col || !round_rows || !steps_rc || n_rounds < 0 || panel_rows < 1 || n_panels < 1) return -1;
  parallel_rounds(n_rounds, n_threads, [&](int32_t r0, int32_t r1) {
    std::vector<int32_t> cnt(n_panels), odd(n_panels);
    for (int32_t r = r0; r < r1; ++r) {
      int32_t *dst = steps_rc + (int64_t)r * n_panels;
      std::fill(dst, dst + n_panels, 0);
      if (round_wide && round_wide[r]) {             // one row over 8 lane groups: 4 take its even panel rows, 4 its odd ones
        const int32_t row = round_rows[(int64_t)r * GROUPS];
        if (row < 0) continue;
        std::fill(cnt.begin(), cnt.end(), 0);
        std::fill(odd.begin(), odd.end(), 0);
        for (int64_t e = rowptr[row]; e < rowptr[row + 1]; ++e) {
          const int32_t c = col[e];
          if (skip_diag && c == row) continue;
          const int32_t p = c / panel_rows;
          ++cnt[p];
          odd[p] += (c - p * panel_rows) & 1;
        }
        for (int32_t p = 0; p < n_panels; ++p) dst[p] = std::max((cnt[p] - odd[p] + 3) / 4, (odd[p] + 3) / 4);
        continue;
      }
      for (int g = 0; g < GROUPS; ++g) {
        const int32_t row = round_rows[(int64_t)r * GROUPS + g];
        if (row < 0) continue;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int64_t e = rowptr[row]; e < rowptr[row + 1]; ++e) {
          const int32_t c = col[e];
          if (skip_diag && c == row) continue;
          ++cnt[c / panel_rows];
        }
        for (int32_t p = 0; p < n_panels; ++p) dst[p] = std::max(dst[p], cnt[p]);
      }
    }
  });
  return 0;
}

// stream (uint16, (total_octs + spare) * 64 values, any content): tile t of (round r, panel p) starts at oct
// tile_oct[r * n_panels + p] and has ceil(steps_rc / 8) octs.  Order inside a (tile, row): the first row of a bank-sharing
// pair takes its entries even, odd, even, ... (panel row parity), the second odd, even, ...; ascending column inside a parity,
// what is left of the longer parity follows; a slot without entry reads the zero row (panel_rows or panel_rows + 1) of the
// parity its partner does not use.
int ggad_spmm_panel_fill(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                         const int32_t *round_wide, int32_t skip_diag, int32_t panel_rows, int32_t n_panels, const int32_t *steps_rc,
                         const int64_t *tile_oct, uint16_t *stream, int64_t total_octs, int32_t spare_octs, int32_t n_threads) {
  if (!rowptr || !col || !round_rows || !steps_rc || !tile_oct || !stream || panel_rows < 2 || panel_rows > 65000) return -1;
  const uint16_t zero_even = (uint16_t)(panel_rows + (panel_rows & 1)), zero_odd = (uint16_t)(panel_rows + 1 - (panel_rows & 1));
  parallel_rounds(n_rounds, n_threads, [&](int32_t r0, int32_t r1) {
    int64_t pos[GROUPS];
    int32_t len[GROUPS];
    for (int32_t r = r0; r < r1; ++r) {
      const int32_t *rows = round_rows + (int64_t)r * GROUPS;
      for (int g = 0; g < GROUPS; ++g) pos[g] = rows[g] >= 0 ? rowptr[rows[g]] : 0;
      for (int32_t p = 0; p < n_panels; ++p) {
        const int32_t steps = steps_rc[(int64_t)r * n_panels + p];
        if (steps == 0) continue;                    // (no row of the round has an entry here; the cursors do not move)
        uint16_t *tile = stream + tile_oct[(int64_t)r * n_panels + p] * 64;
        const int32_t slots = (steps + 7) / 8 * 8;
        const int32_t lo = p * panel_rows, hi = lo + panel_rows;
        if (round_wide && round_wide[r]) {
          const int32_t row = rows[0];
          const int64_t end = rowptr[row + 1];
          int32_t n_even = 0, n_odd = 0;
          int64_t e = pos[0];
          for (; e < end && col[e] < hi; ++e) {
            if (skip_diag && col[e] == row) continue;
            const int32_t local = col[e] - lo;
            if (local & 1) { tile[(int64_t)((n_odd >> 2) >> 3) * 64 + ODD_GROUPS[n_odd & 3] * 8 + ((n_odd >> 2) & 7)] = (uint16_t)local; ++n_odd; }
            else { tile[(int64_t)((n_even >> 2) >> 3) * 64 + EVEN_GROUPS[n_even & 3] * 8 + ((n_even >> 2) & 7)] = (uint16_t)local; ++n_even; }
          }
          pos[0] = e;
          for (int u = 0; u < 4; ++u) {              // padding: the zero row of the group's own parity class
            for (int32_t t = (n_even - u + 3) / 4; t < slots; ++t) tile[(int64_t)(t >> 3) * 64 + EVEN_GROUPS[u] * 8 + (t & 7)] = zero_even;
            for (int32_t t = (n_odd - u + 3) / 4; t < slots; ++t) tile[(int64_t)(t >> 3) * 64 + ODD_GROUPS[u] * 8 + (t & 7)] = zero_odd;
          }
          continue;
        }
        for (int g = 0; g < GROUPS; ++g) {
          len[g] = 0;
          const int32_t row = rows[g];
          if (row < 0) continue;
          const int64_t end = rowptr[row + 1];
          int64_t e = pos[g];
          int32_t n_par[2] = {0, 0};
          for (; e < end && col[e] < hi; ++e)
            if (!(skip_diag && col[e] == row)) ++n_par[(col[e] - lo) & 1];
          const int32_t mn = std::min(n_par[0], n_par[1]);
          const int first = FIRST_ODD[g] ? 1 : 0;
          int32_t i_par[2] = {0, 0};
          for (int64_t f = pos[g]; f < e; ++f) {
            if (skip_diag && col[f] == row) continue;
            const int32_t local = col[f] - lo;
            const int par = local & 1;
            const int32_t i = i_par[par]++;
            const int32_t t = par == first ? i + std::min(i, mn) : i + std::min(i + 1, mn);
            tile[(int64_t)(t >> 3) * 64 + g * 8 + (t & 7)] = (uint16_t)local;
          }
          len[g] = n_par[0] + n_par[1];
          pos[g] = e;
        }
        for (int g = 0; g < GROUPS; ++g) {           // padding: the zero row of the parity the partner's entry does not have
          const int q = PARTNER[g];
          for (int32_t t = len[g]; t < slots; ++t) {
            const int64_t at = (int64_t)(t >> 3) * 64 + (t & 7);
            bool partner_odd = !FIRST_ODD[g];        // both padded: first rows read the even zero row, second rows the odd one
            if (t < len[q]) partner_odd = tile[at + q * 8] & 1;
            tile[at + g * 8] = partner_odd ? zero_even : zero_odd;
          }
        }
      }
    }
  });
  for (int64_t i = total_octs * 64; i < (total_octs + spare_octs) * 64; ++i) stream[i] = zero_even;
  return 0;
}

// 1 when every off-diagonal value equals (float)(r[row] * r[col]) to a relative rtol, else 0  (Csr.value_factors).
int ggad_spmm_panel_values_factor(const int64_t *rowptr, const int32_t *col, const float *val, const double *r, int32_t n_rows,
                                  double rtol, int32_t n_threads) {
  if (!rowptr || !col || !val || !r || n_rows < 0) return -1;
  int ok = 1;
  parallel_rounds(n_rows, n_threads, [&](int32_t a, int32_t b) {
    for (int32_t i = a; i < b; ++i) {
      const double ri = r[i];
      for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        if (col[e] == i) continue;
        const float want = (float)(ri * r[col[e]]);
        const double d = (double)val[e] - (double)want;
        if ((d < 0 ? -d : d) > rtol * (want < 0 ? -(double)want : (double)want)) { __atomic_store_n(&ok, 0, __ATOMIC_RELAXED); return; }
      }
    }
  });
  return ok;
}

}  // extern "C"
