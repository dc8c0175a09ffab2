// Test helper (CPU): read a matrix text file with the host program's reader and write it back with its writer.
//   host_io_roundtrip <in> <out> [digits]
#include "../eigen-lstm_amd/host/matrix_io.h"

#include <cstdio>
#include <cstdlib>
#include <map>

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::map<std::pair<size_t, size_t>, float> m;
    size_t rows = 0, cols = 0;
    if (!matrix_io::read_matrix(argv[1], [&](size_t r, size_t c, double v) { m[{r, c}] = (float)v; }, &rows, &cols)) return 3;
    const int digits = argc > 3 ? atoi(argv[3]) : 6;
    if (!matrix_io::write_matrix(argv[2], rows, cols, [&](size_t r, size_t c) { return m[{r, c}]; }, digits)) return 4;
    printf("%zu %zu\n", rows, cols);
    return 0;
}
