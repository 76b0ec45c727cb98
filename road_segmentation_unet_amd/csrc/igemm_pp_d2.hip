// igemm_pp for dilation 2 (the dilated twin blocks of unet.py:32-39: conv_dilut_*/atrous_conv1|2, forward and backward-data): the same source,
// built with the halo tile 4 pixels wider / higher and the taps 2 pixels apart; its kernels carry DIL = 2 in their names, its two entry
// points the names below. Round 3 measured these instantiations 8-10 % faster per layer than igemm_fwd2's dilated launches and dropped them for
// lack of an effect on config 4's step; config 3 (one patch per step, dilated layers = half of the encoder) is where they count.
#define PP_DIL 2
#define igemm_pp_supports igemm_pp_d2_supports
#define igemm_pp_launch igemm_pp_d2_launch
#include "igemm_pp.hip"
