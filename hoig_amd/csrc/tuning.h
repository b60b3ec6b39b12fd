// Kernel-variant choices (hoig_set_tuning, include/hoig_kernels.h): read by the launchers.
#pragma once
enum HoigTuning { HOIG_TUNE_MFMA16 = 0, HOIG_TUNE_WGRAD16, HOIG_TUNE_IGEMM16, HOIG_TUNE_S2_16, HOIG_TUNE_FLAT5, HOIG_TUNE_FEW128, HOIG_TUNE_WFLAT5, HOIG_TUNE_WGRAD_FEW, HOIG_TUNE_HEAD16, HOIG_TUNE_ADAM_PACK, HOIG_TUNE_D_EARLY, HOIG_TUNE_SPLIT_GRADS, HOIG_TUNE_WGRAD_KO, HOIG_TUNE_COUNT };
int hoig_tuning(int id);
