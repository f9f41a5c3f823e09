// GLV fixed-base MSM kernels for window width 16 (k_msm_glv.inc)
#define GLV_WIDTH 16
#include "k_msm_glv.inc"
