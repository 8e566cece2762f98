// Rectangular linear sum assignment on the HOST ("C++ LAPJV in the extension"): the Hungarian matching of
// open_set/assigners/mask_hungarian_assigner.py:126-131, which the reference solves with scipy.optimize.linear_sum_assignment.
// scipy's solver is the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
// algorithms" (IEEE TAES 2016); this is an independent implementation of that published algorithm with the same scan
// order (unvisited columns initially in DESCENDING index order) and the same tie rule (among equal reduced costs an
// unassigned column wins, otherwise the first one scanned), so that the returned indices -- not just the optimal cost --
// equal scipy's (tests/test_host_logic.py checks this on random, tied and degenerate matrices). All (layer x image)
// problems of a training step are solved by one call, without the Python / numpy round trip per problem.
#include "cgg_common.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

namespace {

// cost: nr x nc row-major with nr <= nc. On success col4row[i] = column assigned to row i.
bool lsap_solve(int nr, int nc, const double* cost, std::vector<int>& col4row) {
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), dist(nc);
  std::vector<int> path(nc, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  col4row.assign(nr, -1);
  for (int cur = 0; cur < nr; ++cur) {
    // shortest augmenting path from row `cur` to any unassigned column
    double min_val = 0.0;
    int n_rem = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(dist.begin(), dist.end(), INF);
    int sink = -1, i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = INF;
      SR[i] = 1;
      for (int it = 0; it < n_rem; ++it) {
        const int j = remaining[it];
        const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
        if (r < dist[j]) {
          path[j] = i;
          dist[j] = r;
        }
        if (dist[j] < lowest || (dist[j] == lowest && row4col[j] == -1)) {
          lowest = dist[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == INF) return false;          // infeasible
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j;
      else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--n_rem];
    }
    // dual update
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - dist[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - dist[j];
    // augment along the path
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      std::swap(col4row[r], j);
      if (r == cur) break;
    }
  }
  return true;
}

int lsap_one(const float* cost, int nr, int nc, int64_t* rows, int64_t* cols) {
  const int n = std::min(nr, nc);
  if (n == 0) return CGG_OK;
  std::vector<double> c((size_t)nr * nc);
  const bool transpose = nc < nr;
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) {
      const double x = (double)cost[(size_t)i * nc + j];
      if (std::isnan(x) || x == -std::numeric_limits<double>::infinity()) return CGG_EINVAL;
      if (transpose) c[(size_t)j * nr + i] = x;
      else c[(size_t)i * nc + j] = x;
    }
  std::vector<int> col4row;
  if (!lsap_solve(transpose ? nc : nr, transpose ? nr : nc, c.data(), col4row)) return CGG_EUNSUPPORTED;
  if (transpose) {
    // solved on the transpose: col4row[j] = original row of original column j; report sorted by original row
    std::vector<int> order(col4row.size());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
    for (size_t k = 0; k < order.size(); ++k) {
      rows[k] = col4row[order[k]];
      cols[k] = order[k];
    }
  } else {
    for (int i = 0; i < nr; ++i) {
      rows[i] = i;
      cols[i] = col4row[i];
    }
  }
  return CGG_OK;
}

}  // namespace

// n_problems cost matrices, f32 row-major, concatenated in `cost`; problem p is nr[p] x nc[p]. rows / cols receive
// min(nr[p], nc[p]) index pairs per problem, concatenated (row indices ascending, as scipy returns them).
extern "C" int cgg_linear_sum_assignment_f32(const float* cost, int n_problems, const int* nr, const int* nc, int64_t* rows,
                                             int64_t* cols) {
  CGG_REQUIRE(n_problems >= 0 && (n_problems == 0 || (cost && nr && nc && rows && cols)), CGG_EINVAL,
              "cgg_linear_sum_assignment_f32: null pointer");
  size_t coff = 0, ooff = 0;
  for (int p = 0; p < n_problems; ++p) {
    CGG_REQUIRE(nr[p] >= 0 && nc[p] >= 0, CGG_EINVAL, "cgg_linear_sum_assignment_f32: problem %d is %d x %d", p, nr[p], nc[p]);
    const int rc = lsap_one(cost + coff, nr[p], nc[p], rows + ooff, cols + ooff);
    CGG_REQUIRE(rc != CGG_EINVAL, CGG_EINVAL, "cgg_linear_sum_assignment_f32: problem %d: matrix contains invalid numeric entries", p);
    CGG_REQUIRE(rc == CGG_OK, CGG_EUNSUPPORTED, "cgg_linear_sum_assignment_f32: problem %d: cost matrix is infeasible", p);
    coff += (size_t)nr[p] * nc[p];
    ooff += (size_t)std::min(nr[p], nc[p]);
  }
  return CGG_OK;
}
