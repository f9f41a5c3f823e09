// GLV fixed-base MSM kernels for window width 14 (k_msm_glv.inc)
#define GLV_WIDTH 14
#include "k_msm_glv.inc"
