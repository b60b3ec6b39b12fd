// Kernel-variant choices (hoig_set_tuning, include/hoig_kernels.h): read by the launchers.
#pragma once
enum HoigTuning { HOIG_TUNE_IGEMM16 = 0, HOIG_TUNE_S2_16, HOIG_TUNE_FLAT5, HOIG_TUNE_WFLAT5, HOIG_TUNE_HEAD16, HOIG_TUNE_D_EARLY, HOIG_TUNE_SPLIT_GRADS, HOIG_TUNE_PAIR, HOIG_TUNE_WDMA16, HOIG_TUNE_S2_PIPE, HOIG_TUNE_NORM_IN, HOIG_TUNE_WINO8, HOIG_TUNE_COUNT };
int hoig_tuning(int id);
