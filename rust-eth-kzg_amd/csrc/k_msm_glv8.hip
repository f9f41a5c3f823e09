// GLV fixed-base MSM kernels for window width 8 (k_msm_glv.inc)
#define GLV_WIDTH 8
#include "k_msm_glv.inc"
