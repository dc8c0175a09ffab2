// matrix_io.h -- matrix text files in the layout the reference writes and reads.
//
// The reference saves a matrix with `file << m` (OV/lstm_eigen_class_CUDA/io.h:16-30), i.e. Eigen's default IOFormat:
// stream precision (6 significant digits), every coefficient right-aligned to ONE common width (the widest entry of
// the whole matrix), columns separated by a single space, rows by '\n', and NO newline after the last row.  Its reader
// (io.h:36-74, readMatrix) loops `while (!infile.eof()) { getline; ...; row++; }`, so a trailing newline makes it touch
// row `rows` (an assertion abort in the reference's build, an out-of-bounds write with NDEBUG): files written here must
// end without one to be loadable by the reference, and they do.  `digits` = 9 round-trips a float exactly (used for
// the Adagrad memory of a resumable checkpoint, which the reference does not save at all).
#pragma once
#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace matrix_io {

// get(r, c) -> value; returns false when the file cannot be opened
template <class Get> bool write_matrix(const std::string &path, size_t rows, size_t cols, Get get, int digits = 6) {
    std::vector<std::string> cells(rows * cols);
    size_t width = 0;
    for (size_t r = 0; r < rows; r++)
        for (size_t c = 0; c < cols; c++) {
            std::ostringstream ss;
            ss.precision(digits);
            ss << get(r, c);
            width = std::max(width, ss.str().size());
            cells[r * cols + c] = ss.str();
        }
    std::ofstream f(path);
    if (!f) return false;
    for (size_t r = 0; r < rows; r++) {
        for (size_t c = 0; c < cols; c++) {
            const std::string &s = cells[r * cols + c];
            f << (c ? " " : "") << std::string(width - s.size(), ' ') << s;
        }
        if (r + 1 < rows) f << "\n";
    }
    return (bool)f;
}

// set(r, c, value) for every entry found; *rows_out / *cols_out = what the file held.  Blank lines are skipped
// (files from older versions of this program ended with a newline).  Returns false if the file cannot be opened or
// its rows have different lengths.
template <class Set> bool read_matrix(const std::string &path, Set set, size_t *rows_out, size_t *cols_out) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    size_t r = 0, cols = 0;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        double v;
        size_t c = 0;
        while (ss >> v) set(r, c++, v);
        if (c == 0) continue;
        if (r > 0 && c != cols) return false;
        cols = c;
        r++;
    }
    *rows_out = r;
    *cols_out = cols;
    return true;
}

} // namespace matrix_io
