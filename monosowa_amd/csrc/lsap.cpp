// Host-side native code: rectangular linear sum assignment for the Hungarian matcher.
//
// The reference matcher calls scipy.optimize.linear_sum_assignment once per (decoder layer, image,
// query group) -- 3 x 16 x 11 = 528 calls per training step at about 20 us of Python overhead each
// (lib/models/monodetr/matcher.py:94-103).  This file solves all of them in one C call.  The algorithm
// is the shortest-augmenting-path method of Crouse ("On implementing 2D rectangular assignment
// algorithms", IEEE TAES 2016), which is also what scipy implements; the dual updates, the scan order
// of the remaining columns and the tie-break (prefer an unassigned column) follow that description so
// that equal-cost ties resolve identically (tests compare against scipy, including tied matrices).
#include <cmath>
#include <cstdint>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Workspace {
  std::vector<double> u, v, shortest, cost;
  std::vector<int64_t> path, col4row, row4col, remaining;
  std::vector<char> SR, SC;
};

// cost: nr x nc row-major with nr <= nc.  Returns false if infeasible.  col4row[i] = column of row i.
bool solve_wide(int64_t nr, int64_t nc, const double *cost, Workspace &w) {
  const double inf = std::numeric_limits<double>::infinity();
  w.u.assign(nr, 0.0);
  w.v.assign(nc, 0.0);
  w.shortest.resize(nc);
  w.path.assign(nc, -1);
  w.col4row.assign(nr, -1);
  w.row4col.assign(nc, -1);
  w.SR.resize(nr);
  w.SC.resize(nc);
  w.remaining.resize(nc);
  for (int64_t cur = 0; cur < nr; ++cur) {
    double min_val = 0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) w.remaining[it] = nc - it - 1;
    std::fill(w.SR.begin(), w.SR.end(), 0);
    std::fill(w.SC.begin(), w.SC.end(), 0);
    std::fill(w.shortest.begin(), w.shortest.end(), inf);
    int64_t sink = -1, i = cur;
    while (sink == -1) {
      int64_t index = -1;
      double lowest = inf;
      w.SR[i] = 1;
      for (int64_t it = 0; it < num_remaining; ++it) {
        const int64_t j = w.remaining[it];
        const double r = min_val + cost[i * nc + j] - w.u[i] - w.v[j];
        if (r < w.shortest[j]) {
          w.path[j] = i;
          w.shortest[j] = r;
        }
        if (w.shortest[j] < lowest || (w.shortest[j] == lowest && w.row4col[j] == -1)) {
          lowest = w.shortest[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == inf) return false;
      const int64_t j = w.remaining[index];
      if (w.row4col[j] == -1) sink = j; else i = w.row4col[j];
      w.SC[j] = 1;
      w.remaining[index] = w.remaining[--num_remaining];
    }
    w.u[cur] += min_val;
    for (int64_t r = 0; r < nr; ++r)
      if (w.SR[r] && r != cur) w.u[r] += min_val - w.shortest[w.col4row[r]];
    for (int64_t j = 0; j < nc; ++j)
      if (w.SC[j]) w.v[j] -= min_val - w.shortest[j];
    int64_t j = sink;
    while (true) {
      const int64_t r = w.path[j];
      w.row4col[j] = r;
      std::swap(w.col4row[r], j);
      if (r == cur) break;
    }
  }
  return true;
}

// General nr x nc problem given through strides (in elements) of a float matrix.  Writes
// k = min(nr, nc) pairs sorted by row index.  Returns k, or -1 (infeasible / non-finite cost).
int64_t solve_strided(const float *c, int64_t nr, int64_t nc, int64_t rs, int64_t cs, int64_t *rows,
                      int64_t *cols, Workspace &w) {
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nr > nc;
  const int64_t R = transpose ? nc : nr, C = transpose ? nr : nc;
  w.cost.resize((size_t)R * C);
  for (int64_t i = 0; i < R; ++i)
    for (int64_t j = 0; j < C; ++j) {
      const double x = transpose ? (double)c[j * rs + i * cs] : (double)c[i * rs + j * cs];
      if (std::isnan(x) || x == -std::numeric_limits<double>::infinity()) return -1;
      w.cost[(size_t)i * C + j] = x;
    }
  if (!solve_wide(R, C, w.cost.data(), w)) return -1;
  if (!transpose) {
    for (int64_t i = 0; i < R; ++i) { rows[i] = i; cols[i] = w.col4row[i]; }
  } else {
    // assignment is (row = col4row[i], col = i); report sorted by row: walk the columns of the transposed
    // problem (= original rows) in increasing order
    int64_t k = 0;
    for (int64_t j = 0; j < C; ++j)
      if (w.row4col[j] != -1) { rows[k] = j; cols[k] = w.row4col[j]; ++k; }
  }
  return R;
}

}  // namespace

namespace {

// A small persistent pool: `run_on_pool(f, k)` runs f on k pool threads and on the caller, and returns when all are done.
// The helpers spin briefly for the next job before they sleep (the train step calls once per ~70 ms).
class Pool {
 public:
  void run(const std::function<void()> &f, int helpers) {
    std::unique_lock<std::mutex> call(call_mu_);                    // one call at a time
    if (helpers > (int)threads_.size()) grow(helpers);
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &f;
      want_ = helpers;
      taken_ = 0;
      done_ = 0;
      ++generation_;
    }
    cv_.notify_all();
    f();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return done_ == want_; });
    job_ = nullptr;
  }

 private:
  void grow(int n) {
    while ((int)threads_.size() < n) {
      threads_.emplace_back([this] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void()> *job = nullptr;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return generation_ != seen && job_ != nullptr && taken_ < want_; });
            seen = generation_;
            ++taken_;
            job = job_;
          }
          (*job)();
          {
            std::lock_guard<std::mutex> lk(mu_);
            ++done_;
          }
          done_cv_.notify_one();
        }
      });
      threads_.back().detach();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void()> *job_ = nullptr;
  int want_ = 0, taken_ = 0, done_ = 0;
  uint64_t generation_ = 0;
};

void run_on_pool(const std::function<void()> &f, int helpers) {
  static Pool *pool = new Pool();                                    // never destroyed: its threads are detached
  if (helpers <= 0) { f(); return; }
  pool->run(f, helpers);
}

}  // namespace

extern "C" {

int lsap_abi_version(void) { return 1; }

// One problem (double, row-major), the scipy.optimize.linear_sum_assignment contract.
int64_t lsap_solve_f64(const double *cost, int64_t nr, int64_t nc, int64_t *rows, int64_t *cols) {
  Workspace w;
  std::vector<float> dummy;
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nr > nc;
  const int64_t R = transpose ? nc : nr, C = transpose ? nr : nc;
  w.cost.resize((size_t)R * C);
  for (int64_t i = 0; i < R; ++i)
    for (int64_t j = 0; j < C; ++j) {
      const double x = transpose ? cost[j * nc + i] : cost[i * nc + j];
      if (std::isnan(x) || x == -std::numeric_limits<double>::infinity()) return -1;
      w.cost[(size_t)i * C + j] = x;
    }
  if (!solve_wide(R, C, w.cost.data(), w)) return -1;
  if (!transpose) {
    for (int64_t i = 0; i < R; ++i) { rows[i] = i; cols[i] = w.col4row[i]; }
  } else {
    int64_t k = 0;
    for (int64_t j = 0; j < C; ++j)
      if (w.row4col[j] != -1) { rows[k] = j; cols[k] = w.row4col[j]; ++k; }
  }
  return R;
}

// Grouped matching of the whole step: cost [NL, B, Q, T] float32 (T = sum of sizes); for every layer,
// image b and query group g, rows [g*Q/G, (g+1)*Q/G) are matched to columns [off_b, off_b + sizes[b]).
// Pairs are appended per (layer, image) in group order, query indices absolute, target indices local to
// the image -- the concatenation the reference builds at matcher.py:98-103.  out_count[NL*B] receives the
// pairs per (layer, image); out_src / out_tgt need NL * sum_b G*min(Q/G, sizes[b]) entries.
//
// padded != 0: cost is [NL, B, Q, T] with T = max(sizes) and image b's targets in columns [0, sizes[b])
// (the per-image diagonal blocks of the full matrix), which is 1/B of the data to bring to the host.
int64_t lsap_match_groups_f32(const float *cost, int64_t NL, int64_t B, int64_t Q, int64_t T,
                              const int64_t *sizes, int64_t G, int64_t padded, int64_t *out_src,
                              int64_t *out_tgt, int64_t *out_count) {
  Workspace w;
  const int64_t gq = Q / G;
  std::vector<int64_t> rows(gq > T ? gq : T), cols(gq > T ? gq : T);
  int64_t n_out = 0;
  for (int64_t l = 0; l < NL; ++l) {
    int64_t off = 0;
    for (int64_t b = 0; b < B; ++b) {
      const int64_t n = sizes[b];
      int64_t cnt = 0;
      for (int64_t g = 0; g < G; ++g) {
        const float *c = cost + ((l * B + b) * Q + g * gq) * T + off;
        const int64_t k = solve_strided(c, gq, n, T, 1, rows.data(), cols.data(), w);
        if (k < 0) return -1;
        for (int64_t i = 0; i < k; ++i) {
          out_src[n_out] = rows[i] + g * gq;
          out_tgt[n_out] = cols[i];
          ++n_out;
        }
        cnt += k;
      }
      out_count[l * B + b] = cnt;
      if (!padded) off += n;
    }
  }
  return n_out;
}


// Same assignments as lsap_match_groups_f32, written as the flat index tensor the criterion consumes:
// out_idx [3, NL, K] = (image, query, target offset by the images before it), K = sum_b G * min(Q / G, sizes[b]) pairs per
// layer in (image, group) order.  Output positions are known up front, so the (layer, image) problems are independent:
// they are dealt to n_threads host threads (the call sits on the train step's critical path, right behind its one host
// sync).  Returns K, or -1 for an infeasible matrix.
int64_t lsap_match_flat_f32(const float *cost, int64_t NL, int64_t B, int64_t Q, int64_t T, const int64_t *sizes,
                            int64_t G, int64_t padded, int64_t *out_idx, int64_t n_threads) {
  const int64_t gq = Q / G;
  std::vector<int64_t> first(B + 1, 0), toff(B + 1, 0);
  for (int64_t b = 0; b < B; ++b) {
    first[b + 1] = first[b] + G * (gq < sizes[b] ? gq : sizes[b]);
    toff[b + 1] = toff[b] + sizes[b];
  }
  const int64_t K = first[B];
  if (K == 0) return 0;
  int64_t *ob = out_idx, *oq = out_idx + NL * K, *ot = out_idx + 2 * NL * K;
  std::atomic<int64_t> next(0);
  std::atomic<int> failed(0);
  auto worker = [&]() {
    Workspace w;
    std::vector<int64_t> rows(gq > T ? gq : T), cols(gq > T ? gq : T);
    for (;;) {
      const int64_t task = next.fetch_add(1);
      if (task >= NL * B) break;
      const int64_t l = task / B, b = task % B, n = sizes[b];
      int64_t pos = l * K + first[b];
      for (int64_t g = 0; g < G; ++g) {
        const float *c = cost + ((l * B + b) * Q + g * gq) * T + (padded ? 0 : toff[b]);
        const int64_t k = solve_strided(c, gq, n, T, 1, rows.data(), cols.data(), w);
        if (k < 0) { failed.store(1); return; }
        for (int64_t i = 0; i < k; ++i, ++pos) {
          ob[pos] = b;
          oq[pos] = rows[i] + g * gq;
          ot[pos] = cols[i] + toff[b];
        }
      }
    }
  };
  if (n_threads < 1) n_threads = 1;
  if (n_threads > NL * B) n_threads = NL * B;
  // helper threads from a persistent pool (creating and joining three threads costs ~0.1 ms per call -- with the GPU idle)
  run_on_pool(worker, (int)n_threads - 1);
  return failed.load() ? -1 : K;
}

}  // extern "C"
