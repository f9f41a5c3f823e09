// GLV fixed-base MSM kernels for window width 15 (k_msm_glv.inc)
#define GLV_WIDTH 15
#include "k_msm_glv.inc"
