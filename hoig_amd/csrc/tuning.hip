// hoig_set_tuning: the one table of kernel-variant choices (see include/hoig_kernels.h).
#include <cstring>
#include "hoig_kernels.h"
#include "tuning.h"

namespace {
struct Entry {
    const char *key;
    int value;
};
Entry g_table[HOIG_TUNE_COUNT] = {
    {"igemm16", 1},
    {"s2_16", 1},
    {"flat5", 2},
    {"wflat5", 1},
    {"head16", 1},
    {"d_early", 1},
    {"split_grads", 1},
    {"pair", 2},
    {"wdma16", 2},
    {"s2_pipe", 1},
    {"norm_in", 1},
    {"wino8", 0},
};
}  // namespace

int hoig_tuning(int id) { return (id >= 0 && id < HOIG_TUNE_COUNT) ? g_table[id].value : 0; }

extern "C" int hoig_set_tuning(const char *key, int value) {
    if (!key) return -1;
    for (int i = 0; i < HOIG_TUNE_COUNT; ++i)
        if (strcmp(g_table[i].key, key) == 0) {
            const int prev = g_table[i].value;
            if (value >= 0) g_table[i].value = value;
            return prev;
        }
    return -1;
}
