// GLV fixed-base MSM kernels for window width 12 (k_msm_glv.inc)
#define GLV_WIDTH 12
#include "k_msm_glv.inc"
