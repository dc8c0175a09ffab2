// rng.h -- the build's seeded replacement for the reference's per-call
// `std::random_device rd; std::mt19937 mt(rd()); std::normal_distribution<> randn(mean, stddev)`
// (R/lstm.cc:370-372, 309-311), which is unseeded and therefore irreproducible.
// Spec (shared with eigen-lstm_amd/lstm_hip.py, restated independently by the test oracle):
//   MT19937 with init_genrand(seed); uniforms = genrand_res53; normals = Marsaglia polar method,
//   the second value of each accepted pair cached.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

class SeededRng {
  public:
    explicit SeededRng(uint32_t seed) {
        mt_[0] = seed;
        for (int i = 1; i < 624; i++) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + (uint32_t)i;
        idx_ = 624;
    }
    uint32_t u32() {
        if (idx_ >= 624) twist();
        uint32_t y = mt_[idx_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double uniform() {
        const uint32_t a = u32() >> 5, b = u32() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
    double normal() {
        if (have_spare_) {
            have_spare_ = false;
            return spare_;
        }
        double u, v, s;
        do {
            u = 2.0 * uniform() - 1.0;
            v = 2.0 * uniform() - 1.0;
            s = u * u + v * v;
        } while (s >= 1.0 || s == 0.0);
        const double m = std::sqrt(-2.0 * std::log(s) / s);
        spare_ = v * m;
        have_spare_ = true;
        return u * m;
    }
    // randn(m, mean, stddev) with the reference's row-outer / column-inner fill of a column-major
    // rows x cols matrix (R/lstm.cc:374-378)
    void randn(float *m, int rows, int cols, double mean, double stddev) {
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) m[(size_t)j * rows + i] = (float)(mean + stddev * normal());
    }

  private:
    void twist() {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (mt_[i] & 0x80000000u) | (mt_[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = mt_[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt_[i] = v;
        }
        idx_ = 0;
    }
    uint32_t mt_[624];
    int idx_;
    bool have_spare_ = false;
    double spare_ = 0.0;
};
