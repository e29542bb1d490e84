// ref_filter_harness.cpp — exposes the REFERENCE's own header-only filters
// (/root/reference/src/cdpr_gazebo/include/cdpr_gazebo/Filter.h, <cmath> only) behind a
// tiny C ABI so tests can pin oracle/cdpr_oracle.c's BiQuad restatement against them.
// Built only where /root/reference exists (see Makefile: target _ref); the header is
// compiled where it lies, never copied.  TEST INFRASTRUCTURE ONLY.
#include "cdpr_gazebo/Filter.h"

extern "C" {
void *ref_biquad_new(double fc, double fs, double q) {
  auto *f = new gazebo::math::BiQuad<double>();
  f->SetValue(0.0);        // as Pid::CascadeFilter's ctor does (Pid.cpp:33-34)
  f->SetFc(fc, fs, q);
  return f;
}
void ref_biquad_set_value(void *p, double v) { static_cast<gazebo::math::BiQuad<double> *>(p)->SetValue(v); }
double ref_biquad_process(void *p, double x) { return static_cast<gazebo::math::BiQuad<double> *>(p)->process(x); }
void ref_biquad_free(void *p) { delete static_cast<gazebo::math::BiQuad<double> *>(p); }

void *ref_onepole_new(double fc, double fs) {
  auto *f = new gazebo::math::OnePole<double>(fc, fs);
  f->SetValue(0.0);
  return f;
}
double ref_onepole_process(void *p, double x) { return static_cast<gazebo::math::OnePole<double> *>(p)->Process(x); }
void ref_onepole_free(void *p) { delete static_cast<gazebo::math::OnePole<double> *>(p); }
}
