// lstm_main.cc -- the C++ host program: the reference's main() (R/lstm.cc:50-361, batched per
// OV/lstm_eigen_opt/lstm.cc:47-413) with its compile-time constants turned into a command line and
// the loop body handed to the MI355X through the C ABI (include/lstm_hip.h).  No HIP, no torch here.
//
//   lstm <text file> <hidden> <seq> <batch> <lr> [options]        (the reference's knobs, R/lstm.cc:53-63)
//   lstm --data F --hidden N --seq S --batch B --lr LR [--epochs E] [--seed K] [--gpus G]
//        [--windows W] [--sample C] [--lr-warmup-windows X] [--save PREFIX] [--load PREFIX]
//        [--eval-file F] [--stride K] [--forget-bias V] [--test-percent F] [--test-every SEC] [--log PREFIX]
//        [--fast-math] [--step-kernels] [--quiet]
//
// stdout follows the reference: "Read N bytes (file)" (R/lstm.cc:398), the carriage-return progress
// line (OV/lstm_eigen_opt/lstm.cc:320-331), the epoch summary (R/lstm.cc:284-291: GFLOP uses 2^30,
// loss divided by S*length) and the "Generated text |...|" block (R/lstm.cc:352-356).
//
// --gpus G forks one process per GPU before anything touches HIP; batch streams are sharded
// rank-major, the RCCL unique id travels over a pipe, and every window ends with one SUM
// all-reduce of the flat gradient block inside the library.
// --test-percent F holds out the last F% of the text (OV/lstm_eigen_class_CUDA/lstm.cc:76-86); --log PREFIX keeps the
// reference's results log (class_CUDA lstm.cc:196-236): every --test-every seconds and at each epoch end a row
// [index, seconds since the last test, train error, test error, GFlOP/s] is appended to PREFIX.txt, the parameters go
// to PREFIX_{W,U,Why,b,by}.txt and 5000 sampled bytes to PREFIX_sample.txt.  --save / --load also carry the Adagrad
// memory (PREFIX_mem_*.txt) and the stream cursors (PREFIX_cursors.txt), which the reference's checkpoints lack.
// --lr-warmup-windows X applies lr = 0 for the first X windows (the reference's GPU driver uses
// X = 50*S, OV/lstm_eigen_class_CUDA/lstm.cc:364-367; 0 = the root file's behaviour).
#include "../../include/lstm_hip.h"
#include "matrix_io.h"
#include "rng.h"

#include <errno.h>
#include <poll.h>
#include <signal.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Options {
    std::string data = "alice29.txt"; // R/lstm.cc:63
    int N = 64, S = 3, B = 1;         // R/lstm.cc:53-57, OV/lstm_eigen_opt/lstm.cc:56
    double lr = 1e-1;                 // R/lstm.cc:59
    long epochs = 1000;               // R/lstm.cc:60
    uint32_t seed = 1;
    int gpus = 1;
    long windows = -1;        // cap on windows per epoch (-1: length - S, R/lstm.cc:151)
    int sample = 1000;        // R/lstm.cc:295
    long lr_warmup = 0;
    int stride = 1;           // bytes per window per stream (lstm_segment.cc: S/2)
    double forget_bias = 0.0; // OV/lstm_eigen_class_batch/lstm.cc:81 uses 1
    std::string save, load, eval_file, log;
    int test_percent = 0;    // last F%% of the text held out (OV/lstm_eigen_class_CUDA/lstm.cc:76-86 uses 1)
    double test_every = 0.0; // seconds between held-out tests + log rows (class_CUDA: 900; 0 = at epoch end only)
    unsigned flags = 0;
    bool quiet = false;
    bool last_step_loss = false; // report forward_loss of OV/lstm_eigen_class_CUDA/lstm.h:200-221 (last step, nats)
    bool last_step_bits = false; // ... or cuLSTM::calculate_loss, cu_lstm.h:203-215 (last step, bits)
};

[[noreturn]] void die(const std::string &m) {
    fprintf(stderr, "lstm: %s\n", m.c_str());
    exit(2);
}
#define CK(call)                                                          \
    do {                                                                  \
        int rc_ = (call);                                                 \
        if (rc_ != 0) die(std::string(#call) + ": " + lstm_hip_last_error()); \
    } while (0)

double now() { // Timer, R/timer.h:21-41
    timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

// rawread, R/lstm.cc:382-420
std::vector<uint8_t> rawread(const std::string &filename) {
    std::vector<uint8_t> v;
    if (FILE *fp = fopen(filename.c_str(), "rb")) {
        char buf[1 << 16];
        while (size_t len = fread(buf, 1, sizeof(buf), fp)) v.insert(v.end(), buf, buf + len);
        fclose(fp);
        if (!v.empty()) printf("Read %zu bytes (%s)\n", v.size(), filename.c_str());
        else printf("Empty file! (%s)\n", filename.c_str());
    } else {
        printf("fopen error: (%s)\n", filename.c_str());
    }
    return v;
}

// count_flops, OV/lstm_eigen_class_CUDA/lstm.cc:722-747 (the model behind the reference's GFlOP/s)
double count_flops(double M, double N, double S, double B) {
    return (S - 1) * ((N * M * B * 2) + (N * 4 * N * B) + (N * 4 * B * 2) + (5 * N * 4 * B) + (6 * N * B) + (M * N * B * 2) +
                      (8 * N * B) + (N * B) + (M * B * N * 3) + (N * B * 6) + (N * M * B * 4) + (N * B * 8) +
                      (N * 4 * B * M * 3) + (N * 4 * B * N * 3) + (N * 4 * B) + (N * 4 * N * B * 2) + (N * B)) +
           8 * (M * N + M + N * 4 * N + N * 4 * M + N * 4);
}

// Parameters::save_to_disk / load_from_disk, OV/lstm_eigen_class_CUDA/lstm.h:83-101, io.h:16-74:
// five text files <prefix>_{W,U,Why,b,by}.txt in the reference's own layout (matrix_io.h): loadable by either program.
struct Block {
    const char *name;
    size_t rows, cols, off;
};
std::vector<Block> blocks(int N, int M) {
    size_t o = 0;
    std::vector<Block> b;
    auto add = [&](const char *n, size_t r, size_t c) {
        b.push_back({n, r, c, o});
        o += r * c;
    };
    add("W", 4 * (size_t)N, M);
    add("U", 4 * (size_t)N, N);
    add("b", 4 * (size_t)N, 1);
    add("Why", M, N);
    add("by", M, 1);
    return b;
}
void save_params(const std::string &prefix, const std::vector<float> &P, int N, int M, int digits = 6) {
    for (const Block &b : blocks(N, M)) {
        const std::string path = prefix + "_" + b.name + ".txt";
        if (!matrix_io::write_matrix(path, b.rows, b.cols, [&](size_t r, size_t c) { return P[b.off + c * b.rows + r]; }, digits))
            die("cannot write " + path);
    }
}
bool load_params(const std::string &prefix, std::vector<float> &P, int N, int M) {
    for (const Block &b : blocks(N, M)) {
        const std::string path = prefix + "_" + b.name + ".txt";
        size_t rows = 0, cols = 0;
        if (!std::ifstream(path).good()) return false;
        if (!matrix_io::read_matrix(path, [&](size_t r, size_t c, double v) {
                if (r < b.rows && c < b.cols) P[b.off + c * b.rows + r] = (float)v;
            }, &rows, &cols))
            die(path + ": rows of different lengths");
        if (rows != b.rows || cols != b.cols)
            die(path + ": " + std::to_string(rows) + " x " + std::to_string(cols) + ", expected " + std::to_string(b.rows) + " x " +
                std::to_string(b.cols));
    }
    return true;
}

// results log, OV/lstm_eigen_class_CUDA/lstm.cc:203-226 + io.h:16-30: the whole 5-column matrix is rewritten after
// every test, in the same layout
void write_results(const std::string &path, const std::vector<std::vector<double>> &rows) {
    printf("Saving a matrix to %s... \n", path.c_str());
    if (!matrix_io::write_matrix(path, rows.size(), rows.empty() ? 0 : rows[0].size(),
                                 [&](size_t r, size_t c) { return (float)rows[r][c]; }))
        printf("file save error: (%s)\n", path.c_str());
}
void save_cursors(const std::string &prefix, const std::vector<uint64_t> &pos) {
    std::ofstream f(prefix + "_cursors.txt");
    if (!f) die("cannot write " + prefix + "_cursors.txt");
    for (uint64_t v : pos) f << v << "\n";
}
bool load_cursors(const std::string &prefix, std::vector<uint64_t> &pos, size_t length, int S) {
    std::ifstream f(prefix + "_cursors.txt");
    if (!f) return false;
    std::vector<uint64_t> v;
    uint64_t x;
    while (f >> x) v.push_back(x);
    if (v.size() != pos.size()) return false; // written for another batch size: fall back to the spread start
    for (uint64_t c : v)
        if (c < (uint64_t)S || c >= length) return false;
    pos = v;
    return true;
}

Options parse(int argc, char **argv) {
    Options o;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) die("missing value for " + a);
            return argv[++i];
        };
        if (a == "--data") o.data = val();
        else if (a == "--hidden") o.N = atoi(val().c_str());
        else if (a == "--seq") o.S = atoi(val().c_str());
        else if (a == "--batch") o.B = atoi(val().c_str());
        else if (a == "--lr") o.lr = atof(val().c_str());
        else if (a == "--epochs") o.epochs = atol(val().c_str());
        else if (a == "--seed") o.seed = (uint32_t)strtoul(val().c_str(), nullptr, 10);
        else if (a == "--gpus") o.gpus = atoi(val().c_str());
        else if (a == "--windows") o.windows = atol(val().c_str());
        else if (a == "--sample") o.sample = atoi(val().c_str());
        else if (a == "--lr-warmup-windows") o.lr_warmup = atol(val().c_str());
        else if (a == "--stride") o.stride = atoi(val().c_str());
        else if (a == "--forget-bias") o.forget_bias = atof(val().c_str());
        else if (a == "--save") o.save = val();
        else if (a == "--load") o.load = val();
        else if (a == "--eval-file") o.eval_file = val();
        else if (a == "--log") o.log = val();
        else if (a == "--test-percent") o.test_percent = atoi(val().c_str());
        else if (a == "--test-every") o.test_every = atof(val().c_str());
        else if (a == "--fast-math") o.flags |= LSTM_HIP_FAST_MATH;
        else if (a == "--step-kernels") o.flags |= LSTM_HIP_STEP_KERNELS;
        else if (a == "--last-step-loss") o.last_step_loss = true;
        else if (a == "--last-step-loss-bits") o.last_step_bits = true;
        else if (a == "--quiet") o.quiet = true;
        else if (a == "-h" || a == "--help") {
            printf("usage: lstm <text file> <hidden> <seq> <batch> <lr> [--epochs E --seed K --gpus G --windows W --sample C\n"
                   "            --lr-warmup-windows X --save PREFIX --load PREFIX --eval-file F --stride K --forget-bias V\n"
                   "            --test-percent F --test-every SEC --log PREFIX --last-step-loss --last-step-loss-bits --fast-math --step-kernels --quiet]\n");
            exit(0);
        } else if (a.rfind("--", 0) == 0) die("unknown option " + a);
        else pos.push_back(a);
    }
    if (pos.size() > 0) o.data = pos[0];
    if (pos.size() > 1) o.N = atoi(pos[1].c_str());
    if (pos.size() > 2) o.S = atoi(pos[2].c_str());
    if (pos.size() > 3) o.B = atoi(pos[3].c_str());
    if (pos.size() > 4) o.lr = atof(pos[4].c_str());
    if (o.gpus < 1 || o.B % o.gpus != 0) die("--batch must be a multiple of --gpus");
    if (o.test_percent < 0 || o.test_percent > 50) die("--test-percent must be 0..50");
    return o;
}

// one rank = one GPU.  `up`/`down` are pipes to/from the parent when gpus > 1.
int run_rank(const Options &o, int rank, int up, int down) {
    const int M = LSTM_HIP_VOCAB, N = o.N, S = o.S, Bl = o.B / o.gpus;
    const bool lead = rank == 0;
    std::vector<uint8_t> data = rawread(o.data);
    std::vector<uint8_t> testdata;
    if (o.test_percent > 0) { // first (100-F)% trains, the rest is held out (OV/lstm_eigen_class_CUDA/lstm.cc:76-86)
        const size_t percent_size = data.size() / 100, cut = (size_t)(100 - o.test_percent) * percent_size;
        testdata.assign(data.begin() + cut, data.end());
        data.resize(cut);
        if (lead) printf("Train set size: %zu, Test set size: %zu, Total: %zu\n", data.size(), testdata.size(), data.size() + testdata.size());
    }
    if (data.size() <= (size_t)S + 1) die("text too short");
    const size_t length = data.size();

    lstm_hip_config cfg{N, M, S, Bl, rank, o.flags};
    lstm_hip_t *h = nullptr;
    CK(lstm_hip_create(&cfg, &h));
    if (o.gpus > 1) {
        uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES];
        if (lead) {
            CK(lstm_hip_comm_unique_id(id));
            if (write(up, id, sizeof(id)) != (ssize_t)sizeof(id)) die("pipe write");
        }
        if (read(down, id, sizeof(id)) != (ssize_t)sizeof(id)) die("pipe read");
        CK(lstm_hip_comm_init(h, id, o.gpus, rank));
        CK(lstm_hip_set_global_batch(h, o.B));
    }

    // init: W, U, Why ~ N(0, 0.01) in that order, b = by = 0 (R/lstm.cc:113-119); same on all ranks
    const size_t np = lstm_hip_param_count(N, M);
    std::vector<float> P(np, 0.0f);
    SeededRng rng(o.seed);
    {
        auto bl = blocks(N, M);
        rng.randn(P.data() + bl[0].off, 4 * N, M, 0.0, 0.01);
        rng.randn(P.data() + bl[1].off, 4 * N, N, 0.0, 0.01);
        rng.randn(P.data() + bl[3].off, M, N, 0.0, 0.01);
        for (int j = 0; j < N; j++) P[bl[2].off + 2 * N + j] = (float)o.forget_bias; // f-gate bias
    }
    if (!o.load.empty()) {
        if (load_params(o.load, P, N, M)) {
            if (lead) printf("Loaded parameters from %s_{W,U,Why,b,by}.txt\n", o.load.c_str());
        } else if (lead) printf("fopen error: (%s_W.txt) -- keeping the random initialisation\n", o.load.c_str());
    }
    CK(lstm_hip_set_params(h, 0, P.data()));
    if (!o.load.empty()) { // resume: Adagrad memory, when the checkpoint has it
        std::vector<float> mem(np, 0.0f);
        if (std::ifstream(o.load + "_mem_W.txt").good() && load_params(o.load + "_mem", mem, N, M)) {
            CK(lstm_hip_set_params(h, 2, mem.data()));
            if (lead) printf("Loaded Adagrad memory from %s_mem_{W,U,Why,b,by}.txt\n", o.load.c_str());
        }
    }
    CK(lstm_hip_set_text(h, data.data(), length));

    // cursors: deterministic stand-in for rand() % (length - S) + S (OV/lstm_eigen_opt/lstm.cc:140-144)
    std::vector<uint64_t> start_all(o.B), pos(Bl); // start_all: every stream of the global batch, rank-major
    for (int b = 0; b < o.B; b++) start_all[b] = (uint64_t)S + ((uint64_t)b * (length - S)) / (uint64_t)o.B;
    if (!o.load.empty() && load_cursors(o.load, start_all, length, S) && lead) // resume: the streams' positions
        printf("Loaded stream cursors from %s_cursors.txt\n", o.load.c_str());
    std::copy(start_all.begin() + (size_t)rank * Bl, start_all.begin() + (size_t)(rank + 1) * Bl, pos.begin());
    CK(lstm_hip_set_cursors(h, pos.data()));
    CK(lstm_hip_reset_window(h));
    if (o.last_step_loss) CK(lstm_hip_set_loss_mode(h, LSTM_HIP_LOSS_LAST_STEP_NATS));
    if (o.last_step_bits) CK(lstm_hip_set_loss_mode(h, LSTM_HIP_LOSS_LAST_STEP_BITS));
    if (o.stride > 1) CK(lstm_hip_set_stride(h, o.stride, o.stride - 1)); // segment variant: carry from column seg-1

    const double flops_per_iteration = count_flops(M, N, S, o.B);

    // held-out text: --eval-file wins over the --test-percent split
    std::vector<uint8_t> evaldata = (lead && !o.eval_file.empty()) ? rawread(o.eval_file) : testdata;
    const std::string evalname = !o.eval_file.empty() ? o.eval_file : "last " + std::to_string(o.test_percent) + "% of " + o.data;
    std::vector<std::vector<double>> results;
    double last_test = now();
    SeededRng log_rng(o.seed + 7919u); // sampling for the log must not disturb the training stream
    // test(p, testdata) + results row + checkpoint + sample file, OV/lstm_eigen_class_CUDA/lstm.cc:186-236 (lead rank only:
    // the evaluator and the sampler are local to one handle, the other ranks wait at the next all-reduce)
    auto test_and_log = [&](double train_error, double gflops) {
        const double test_time = now() - last_test;
        double test_error = NAN;
        if (evaldata.size() > 1) CK(lstm_hip_eval_bits(h, evaldata.data(), evaldata.size(), &test_error));
        printf("\nTrain error: %g, Test error: %g\n", train_error, test_error);
        if (evaldata.size() > 1) printf("Test error: %.5f bits/char (%s)\n", test_error, evalname.c_str());
        if (!o.log.empty()) {
            results.push_back({(double)results.size(), test_time, train_error, test_error, gflops});
            printf("%6g %6g %6g %6g %6g\n\n", results.back()[0], test_time, train_error, test_error, gflops);
            write_results(o.log + ".txt", results);
            CK(lstm_hip_get_params(h, 0, P.data()));
            save_params(o.log, P, N, M);
            const int n = 5000; // class_CUDA lstm.cc:229
            std::vector<float> h0(N), c0(N);
            log_rng.randn(h0.data(), N, 1, 0.0, 0.1);
            log_rng.randn(c0.data(), N, 1, 0.0, 0.1);
            std::vector<double> u(n);
            for (double &x : u) x = log_rng.uniform();
            std::vector<uint8_t> text(n);
            CK(lstm_hip_sample(h, h0.data(), c0.data(), u.data(), n, text.data()));
            std::ofstream f(o.log + "_sample.txt", std::ios::out | std::ios::binary);
            f.write(reinterpret_cast<const char *>(text.data()), n);
        }
        last_test = now();
    };
    const long windows_per_epoch = (o.windows > 0) ? o.windows : (long)((length - S + o.stride - 1) / o.stride);
    long done_windows = 0;
    std::vector<float> hs((size_t)N * o.B), cs((size_t)N * o.B);
    std::vector<double> losses;

    for (long e = 0; e < o.epochs; e++) {
        // epoch start: h[t], c[t] ~ N(0, 0.1) for every t (OV/lstm_eigen_opt/lstm.cc:176-181); drawn for
        // the GLOBAL batch so every rank consumes the same stream, each keeps its own columns
        for (int t = 0; t < S; t++) {
            rng.randn(hs.data(), N, o.B, 0.0, 0.1);
            rng.randn(cs.data(), N, o.B, 0.0, 0.1);
            CK(lstm_hip_set_state(h, t, hs.data() + (size_t)rank * Bl * N, cs.data() + (size_t)rank * Bl * N));
        }
        double epoch_loss = 0.0;
        long nan_windows = 0;
        const double t0 = now();
        double tf = t0;
        for (long i = 0; i < windows_per_epoch;) {
            long chunk = std::min<long>(100, windows_per_epoch - i); // progress every 100 iterations (opt:320)
            double lr = o.lr;
            if (done_windows < o.lr_warmup) {
                lr = 0.0;
                chunk = std::min<long>(chunk, o.lr_warmup - done_windows);
            }
            losses.resize(chunk);
            CK(lstm_hip_train_windows(h, chunk, lr, losses.data(), nullptr));
            for (double v : losses) {
                if (!std::isnan(v)) epoch_loss += v; // NaN guard as OV/lstm_eigen_class_CUDA/lstm.cc:325-326
                else nan_windows++;
            }
            i += chunk;
            done_windows += chunk;
            // mid-epoch report (class_CUDA lstm.cc:186-188).  With several ranks the lead only has its own share of the loss
            // (local surprisal sum / GLOBAL batch); the ranks exchange sums once per epoch, not here, so the running figure is
            // the lead's share scaled by the rank count -- an estimate over its streams, labelled as what the reference prints
            if (lead && o.test_every > 0 && now() - last_test > o.test_every)
                test_and_log(epoch_loss * (double)o.gpus / ((double)S * (double)i),
                             (i * flops_per_iteration / std::pow(2.0, 30)) / (now() - t0));
            if (lead && !o.quiet) {
                const double t1 = now();
                printf("%9.2f%% %9.2f GFlOP/s\r", 100.0 * (double)(i + S) / (double)length,
                       (chunk * flops_per_iteration / std::pow(2.0, 30)) / (t1 - tf));
                fflush(stdout);
                tf = t1;
            }
        }
        const double epoch_time = now() - t0;
        if (o.gpus > 1) { // sum the ranks' partial losses (each already divided by the global batch)
            if (write(up, &epoch_loss, sizeof(double)) != (ssize_t)sizeof(double)) die("pipe write");
            if (read(down, &epoch_loss, sizeof(double)) != (ssize_t)sizeof(double)) die("pipe read");
        }
        if (lead) {
            const double chars = (double)(S - 1) * o.B * windows_per_epoch;
            printf("\n====================================================================================\n");
            printf("Epoch %ld/%ld, t = %.3f s, est GFLOP/s = %.3f, avg loss = %.3f bits/char\n", e + 1, o.epochs, epoch_time,
                   (flops_per_iteration * windows_per_epoch / std::pow(2.0, 30)) / epoch_time,
                   epoch_loss / ((double)S * (double)(windows_per_epoch + S))); // R/lstm.cc:290: loss/(S*length)
            printf("chars/s through fwd+BPTT = %.1f (%ld windows, %d GPU%s)\n", chars / epoch_time, windows_per_epoch, o.gpus,
                   o.gpus > 1 ? "s" : "");
            if (nan_windows > 0) // the reference skips NaN losses silently; the unshifted softmax (R/lstm.cc:199) overflows when lr is too large
                printf("!!!! %ld of %ld windows had a NaN loss (skipped in the average): lower --lr or use --lr-warmup-windows\n",
                       nan_windows, windows_per_epoch);
            if (evaldata.size() > 1 || !o.log.empty())
                test_and_log(epoch_loss / ((double)S * (double)(windows_per_epoch + S)),
                             (flops_per_iteration * windows_per_epoch / std::pow(2.0, 30)) / epoch_time);
            if (o.sample > 0) { // R/lstm.cc:293-356
                std::vector<float> h0(N), c0(N);
                rng.randn(h0.data(), N, 1, 0.0, 0.1);
                rng.randn(c0.data(), N, 1, 0.0, 0.1);
                std::vector<double> u(o.sample);
                for (double &x : u) x = rng.uniform();
                std::vector<uint8_t> text(o.sample);
                CK(lstm_hip_sample(h, h0.data(), c0.data(), u.data(), o.sample, text.data()));
                printf("\n\n************ Generated text |");
                fwrite(text.data(), 1, text.size(), stdout);
                printf("| Generated text END ************\n");
            } else { // keep the RNG stream aligned across ranks
            }
            if (!o.save.empty()) {
                CK(lstm_hip_get_params(h, 0, P.data()));
                save_params(o.save, P, N, M);
                std::vector<float> mem(np);
                CK(lstm_hip_get_params(h, 2, mem.data()));
                save_params(o.save + "_mem", mem, N, M, 9);
                std::vector<uint64_t> all(o.B); // every stream has advanced by the same number of bytes
                const uint64_t span = (uint64_t)(length - S), adv = (uint64_t)done_windows * (uint64_t)o.stride;
                for (int b = 0; b < o.B; b++) all[b] = (uint64_t)S + ((start_all[b] - S) + adv) % span;
                std::vector<uint64_t> mine(Bl);
                CK(lstm_hip_get_cursors(h, mine.data()));
                for (int b = 0; b < Bl; b++)
                    if (mine[b] != all[(size_t)rank * Bl + b]) die("cursor bookkeeping out of step with the library");
                save_cursors(o.save, all);
                printf("Saved parameters to %s_{W,U,Why,b,by}.txt (+ _mem_*, _cursors)\n", o.save.c_str());
            }
            fflush(stdout);
        }
        if (!lead && o.sample > 0) { // non-lead ranks consume the same draws so later epochs stay in step
            std::vector<float> tmp(N);
            rng.randn(tmp.data(), N, 1, 0.0, 0.1);
            rng.randn(tmp.data(), N, 1, 0.0, 0.1);
            for (int k = 0; k < o.sample; k++) (void)rng.uniform();
        }
    }
    CK(lstm_hip_destroy(h));
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    Options o = parse(argc, argv);
    if (o.gpus == 1) return run_rank(o, 0, -1, -1);

    // one process per GPU, forked before any HIP call
    std::vector<int> up_r(o.gpus), down_w(o.gpus);
    std::vector<pid_t> kids(o.gpus);
    for (int r = 0; r < o.gpus; r++) {
        int up[2], down[2];
        if (pipe(up) != 0 || pipe(down) != 0) die("pipe");
        setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); // RCCL across processes needs dmabuf IPC on this driver
        pid_t pid = fork();
        if (pid < 0) die("fork");
        if (pid == 0) {
            close(up[0]);
            close(down[1]);
            if (r != 0) { // only rank 0 talks on stdout
                if (!freopen("/dev/null", "w", stdout)) _exit(3);
            }
            _exit(run_rank(o, r, up[1], down[0]));
        }
        close(up[1]);
        close(down[0]);
        up_r[r] = up[0];
        down_w[r] = down[1];
        kids[r] = pid;
    }
    signal(SIGPIPE, SIG_IGN); // a rank that died must show up as a failed write, not end the parent
    // relay the unique id, then one loss sum per epoch
    uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES];
    if (read(up_r[0], id, sizeof(id)) != (ssize_t)sizeof(id)) die("rank 0 did not produce a unique id");
    for (int r = 0; r < o.gpus; r++)
        if (write(down_w[r], id, sizeof(id)) != (ssize_t)sizeof(id)) die("relay");
    // The relay never blocks on ONE rank's pipe: a rank that fails (a non-finite window, a hand-off time-out, a HIP error)
    // leaves the others blocked inside the next all-reduce, so its death must be noticed while its peers are silent.
    // poll() over all pipes with a short time-out, children reaped without blocking in between.
    bool relay_failed = false;
    std::vector<bool> done(o.gpus, false);
    int live = o.gpus, rc = 0;
    auto reap_nonblocking = [&]() {
        for (;;) {
            int st = 0;
            const pid_t p = waitpid(-1, &st, WNOHANG);
            if (p <= 0) break;
            for (int r = 0; r < o.gpus; r++)
                if (kids[r] == p && !done[r]) {
                    done[r] = true;
                    live--;
                    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                        rc = 1;
                        relay_failed = true;
                    }
                }
        }
    };
    for (long e = 0; e < o.epochs && !relay_failed; e++) {
        double sum = 0.0;
        std::vector<double> part(o.gpus, 0.0);
        std::vector<bool> got(o.gpus, false);
        int missing = o.gpus;
        while (missing > 0 && !relay_failed) {
            std::vector<pollfd> fds;
            std::vector<int> who;
            for (int r = 0; r < o.gpus; r++)
                if (!got[r]) {
                    fds.push_back(pollfd{up_r[r], POLLIN, 0});
                    who.push_back(r);
                }
            const int n = poll(fds.data(), (nfds_t)fds.size(), 200);
            if (n < 0 && errno != EINTR) relay_failed = true;
            for (size_t k = 0; k < fds.size() && n > 0; k++) {
                if (!(fds[k].revents & (POLLIN | POLLHUP | POLLERR))) continue;
                double v = 0.0;
                if (read(fds[k].fd, &v, sizeof(v)) != (ssize_t)sizeof(v)) { // EOF: that rank is gone
                    relay_failed = true;
                    break;
                }
                part[who[k]] = v;
                got[who[k]] = true;
                missing--;
            }
            reap_nonblocking(); // a child that ended (badly, or before delivering) while its peers are blocked
            if (live < o.gpus && missing > 0) {
                for (int r = 0; r < o.gpus; r++)
                    if (done[r] && !got[r]) relay_failed = true; // it will never deliver
            }
        }
        if (relay_failed) break;
        for (int r = 0; r < o.gpus; r++) sum += part[r]; // rank order: the same sum on every run
        for (int r = 0; r < o.gpus; r++)
            if (write(down_w[r], &sum, sizeof(sum)) != (ssize_t)sizeof(sum)) relay_failed = true;
    }
    // Reap.  A rank that fails (a non-finite window, a hand-off time-out, a HIP error) leaves the others blocked inside the
    // next all-reduce: as soon as the relay breaks or any child ends badly, the remaining children are terminated instead
    // of waited for.
    bool failed = relay_failed;
    if (failed) rc = 1;
    while (live > 0) {
        if (failed)
            for (int r = 0; r < o.gpus; r++)
                if (!done[r]) kill(kids[r], SIGTERM);
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) break;
        for (int r = 0; r < o.gpus; r++)
            if (kids[r] == p && !done[r]) {
                done[r] = true;
                live--;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                    rc = 1;
                    failed = true;
                }
            }
    }
    return rc;
}
