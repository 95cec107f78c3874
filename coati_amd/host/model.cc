// See model.hpp.  fp32 everywhere the reference computes in fp32 (rate
// matrices, marginal sums, logs); only the matrix exponential, for which the
// reference calls the un-vendored Eigen 3.4 `MatrixBase::exp()`
// (mutation_coati.cc:122, mutation_ecm.cc:181), is an own implementation:
// scaling-and-squaring of a degree-18 Taylor polynomial evaluated in fp64 on
// the fp32 input, rounded to fp32 once at the end.  It agrees with the
// reference's golden table mg94P (src/include/coati/mg94p.tcc:26) to fp32
// rounding (tests/test_host_model.py); bit parity with Eigen's fp32 Pade
// evaluation is NOT claimed ("parity unpinned" for expm).
#include "model.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "codon.hpp"

namespace coati_amd {

namespace {

#include "ecm_kosiol2007.inc"

constexpr int N = 61;

float ecm_exchange(int i, int j) {
    if(i == j) return 0.0f;
    const int hi = std::max(i, j), lo = std::min(i, j);
    return kEcmLower[hi * (hi - 1) / 2 + lo];
}

// utils.hpp:134-156 (float overloads)
float log1p_exp(float x) {
    if(x <= -16.0f) return ::expf(x);
    if(x <= 8.0f) return ::log1pf(::expf(x));
    if(x <= 14.5f) return x + ::expf(-x);
    return x;
}
float log_sum_exp(float a, float b) {
    const float hi = std::max(a, b);
    return hi + log1p_exp(-std::fabs(a - b));
}
float log_sum_exp(float a, float b, float c) { return log_sum_exp(log_sum_exp(a, b), c); }

int cod_distance(int c1, int c2) {
    int d = 0;
    for(int p = 0; p < 3; ++p) d += get_nuc(c1, p) != get_nuc(c2, p);
    return d;
}

// Normalise and exponentiate: P = exp(Q * (br_len / d)), fp32 scaling as upstream.
matrix61_t exp_scaled(const matrix61_t& Q, float scale) {
    matrix61_t A(N * N);
    for(int k = 0; k < N * N; ++k) A[k] = Q[k] * scale;
    return expm61(A);
}

}  // namespace

std::array<float, 4> gap_log_consts(const gap_t& gap) {
    if(!(gap.open > 0.0f && gap.open < 1.0f))
        throw std::invalid_argument("Gap opening score must be in range (0,1).");
    if(!(gap.extend > 0.0f && gap.extend < 1.0f))
        throw std::invalid_argument("Gap extension score must be in range (0,1).");
    return {std::log1pf(-gap.open), std::log1pf(-gap.extend), ::logf(gap.open), ::logf(gap.extend)};
}

matrix61_t expm61(const matrix61_t& Af) {
    std::vector<double> A(Af.begin(), Af.end());
    // 1-norm, then scale so that ||A / 2^s||_1 <= 1/4
    double norm = 0.0;
    for(int j = 0; j < N; ++j) {
        double col = 0.0;
        for(int i = 0; i < N; ++i) col += std::fabs(A[i * N + j]);
        norm = std::max(norm, col);
    }
    int squarings = 0;
    if(norm > 0.25) squarings = std::max(0, static_cast<int>(std::ceil(std::log2(norm / 0.25))));
    const double sc = std::ldexp(1.0, -squarings);
    for(double& v : A) v *= sc;

    auto matmul = [](const std::vector<double>& X, const std::vector<double>& Y, std::vector<double>& Z) {
        std::fill(Z.begin(), Z.end(), 0.0);
        for(int i = 0; i < N; ++i)
            for(int k = 0; k < N; ++k) {
                const double x = X[i * N + k];
                if(x == 0.0) continue;
                for(int j = 0; j < N; ++j) Z[i * N + j] += x * Y[k * N + j];
            }
    };
    // Taylor: E = I + A + A^2/2! + ... + A^18/18!
    std::vector<double> E(N * N, 0.0), term(N * N, 0.0), next(N * N);
    for(int i = 0; i < N; ++i) E[i * N + i] = term[i * N + i] = 1.0;
    for(int k = 1; k <= 18; ++k) {
        matmul(term, A, next);
        for(int q = 0; q < N * N; ++q) {
            term[q] = next[q] / k;
            E[q] += term[q];
        }
    }
    for(int s = 0; s < squarings; ++s) {
        matmul(E, E, next);
        E.swap(next);
    }
    return matrix61_t(E.begin(), E.end());
}

std::array<float, 16> gtr_q(const std::array<float, 4>& pi, const std::array<float, 6>& sigma) {
    if(std::any_of(sigma.begin(), sigma.end(), [](float f) { return f < 0.f || f > 1.f; }))
        throw std::invalid_argument("Sigma values must be in range [0,1].");
    std::array<float, 16> q{};
    // sigma order: AC AG AT CG CT GT  (mutation_coati.cc:332-338)
    const int pairs[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for(int s = 0; s < 6; ++s) {
        q[pairs[s][0] * 4 + pairs[s][1]] = sigma[s];
        q[pairs[s][1] * 4 + pairs[s][0]] = sigma[s];
    }
    for(int i = 0; i < 4; ++i)
        for(int j = 0; j < 4; ++j) q[i * 4 + j] *= pi[j];
    for(int i = 0; i < 4; ++i) {
        float off = 0.0f;  // left-to-right sum of the three off-diagonal entries
        bool first = true;
        for(int j = 0; j < 4; ++j) {
            if(j == i) continue;
            off = first ? q[i * 4 + j] : off + q[i * 4 + j];
            first = false;
        }
        q[i * 4 + i] = -off;
    }
    return q;
}

matrix61_t mg94_p(float br_len, float omega, const std::array<float, 4>& nuc_freqs,
                  const std::array<float, 6>& sigma) {
    if(br_len <= 0) throw std::out_of_range("Branch length must be positive.");
    std::array<float, 16> nuc_q;
    if(std::any_of(sigma.begin(), sigma.end(), [](float f) { return f > 0.f; })) {
        nuc_q = gtr_q(nuc_freqs, sigma);
    } else {
        // Yang (1994), "Estimating the pattern of nucleotide substitution", J Mol Evol 39:105
        // (values as used upstream, mutation_coati.cc:65-68)
        nuc_q = {-0.818, 0.132, 0.586, 0.1,      // NOLINT
                 0.221,  -1.349, 0.231, 0.897,   // NOLINT
                 0.909,  0.215, -1.322, 0.198,   // NOLINT
                 0.1,    0.537, 0.128, -0.765};  // NOLINT
    }
    matrix61_t Q(N * N, 0.0f);
    float d = 0.0f;
    for(int i = 0; i < N; ++i) {
        const float pi_i = nuc_freqs[get_nuc(i, 0)] * nuc_freqs[get_nuc(i, 1)] * nuc_freqs[get_nuc(i, 2)];
        float row_sum = 0.0f;
        for(int j = 0; j < N; ++j) {
            float q = 0.0f;
            if(i != j && cod_distance(i, j) == 1) {
                const float w = amino_acid61(i) == amino_acid61(j) ? 1.0f : omega;
                int pos = 0;
                while(get_nuc(i, pos) == get_nuc(j, pos)) ++pos;
                q = w * nuc_q[get_nuc(i, pos) * 4 + get_nuc(j, pos)];
            }
            Q[i * N + j] = q;
            row_sum += q;
        }
        Q[i * N + i] = -row_sum;
        d += pi_i * row_sum;
    }
    return exp_scaled(Q, br_len / d);
}

matrix61_t ecm_p(float br_len, float omega) {
    if(br_len <= 0) throw std::out_of_range("Branch length must be positive.");
    matrix61_t Q(N * N, 0.0f);
    float d = 0.0f;
    for(int i = 0; i < N; ++i) {
        float row_sum = 0.0f;
        for(int j = 0; j < N; ++j) {
            if(i == j) continue;
            // k(i, j, 0) == 1: ts/tv bias is implicit in the exchangeabilities (mutation_ecm.cc:110-113)
            float q = ecm_exchange(i, j) * kEcmFreq[j] * 1.0f;
            if(amino_acid61(i) != amino_acid61(j)) q = q * omega;
            Q[i * N + j] = q;
            row_sum += q;
        }
        Q[i * N + i] = -row_sum;
        d += kEcmFreq[i] * row_sum;
    }
    return exp_scaled(Q, br_len / d);
}

matrix61_t rate_matrix_p(const matrix61_t& Q, float br_len) {
    if(Q.size() != static_cast<std::size_t>(N * N)) throw std::invalid_argument("Rate matrix must be 61x61.");
    return exp_scaled(Q, br_len);
}

table_t marginal_p(const matrix61_t& P, const std::array<float, 4>& pi, AmbiguousNucs amb,
                   MarginalSubst msub) {
    table_t p(kTableRows * kTableCols, 0.0f);
    auto at = [&p](int row, int col) -> float& { return p[row * kTableCols + col]; };
    for(int cod = 0; cod < N; ++cod) {
        for(int nuc = 0; nuc < 4; ++nuc) {
            for(int pos = 0; pos < 3; ++pos) {
                float marg = 0.0f;
                for(int i = 0; i < N; ++i) {
                    const float v = get_nuc(i, pos) == nuc ? P[cod * N + i] : 0.0f;
                    if(msub == MarginalSubst::SUM) {
                        marg += v;
                    } else if(v > marg) {
                        marg = v;
                    }
                }
                at(cod * 3 + pos, nuc) = ::logf(marg / pi[nuc]);
            }
        }
    }
    // IUPAC ambiguity columns: R Y M K S W B D H V N  (mutation_coati.cc:236-306)
    static const int members[11][4] = {{0, 2, -1, -1}, {1, 3, -1, -1}, {0, 1, -1, -1}, {2, 3, -1, -1},
                                       {1, 2, -1, -1}, {0, 3, -1, -1}, {1, 2, 3, -1},  {0, 2, 3, -1},
                                       {0, 1, 3, -1},  {0, 1, 2, -1},  {0, 1, 2, 3}};
    for(int row = 0; row < static_cast<int>(kTableRows); ++row) {
        for(int k = 0; k < 11; ++k) {
            const int* m = members[k];
            float v;
            if(amb == AmbiguousNucs::SUM) {
                if(m[2] < 0) {
                    v = log_sum_exp(at(row, m[0]), at(row, m[1]));
                } else if(m[3] < 0) {
                    v = log_sum_exp(at(row, m[0]), at(row, m[1]), at(row, m[2]));
                } else {
                    v = log_sum_exp(log_sum_exp(at(row, m[0]), at(row, m[1]), at(row, m[2])), at(row, m[3]));
                }
            } else {
                v = std::max(at(row, m[0]), at(row, m[1]));
                if(m[2] >= 0) v = std::max(v, at(row, m[2]));
                if(m[3] >= 0) v = std::max(v, at(row, m[3]));
            }
            at(row, 4 + k) = v;
        }
    }
    return p;
}

table_t set_subst(const model_params_t& prm) {
    if(prm.model == "mar-ecm") return marginal_p(ecm_p(prm.br_len, prm.omega), prm.pi, prm.amb, prm.sub);
    if(prm.model == "mar-mg") return marginal_p(mg94_p(prm.br_len, prm.omega, prm.pi, prm.sigma), prm.pi, prm.amb, prm.sub);
    throw std::invalid_argument("Mutation model unknown.");
}

}  // namespace coati_amd
