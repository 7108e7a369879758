// Library-level entry points and error plumbing of libojf.
#include "ojf_common.h"

namespace ojf {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(const std::string &msg)
{
    g_last_error = msg;
    return -1;
}

int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
}

Camera make_camera(const float *Ki, const float *E, const double *origin, double res)
{
    Camera c;
    for (int i = 0; i < 9; ++i) c.Ki[i] = Ki[i];
    for (int i = 0; i < 12; ++i) c.E[i] = E[i];
    for (int i = 0; i < 3; ++i) {
        c.origin[i] = origin[i];
        // modules/extractor.py:315  eye_v = (eye - origin) / resolution, eye = float32 E[:, :3, 3]
        c.eye_v[i] = ((double)E[4 * i + 3] - origin[i]) / res;
    }
    c.res = res;
    return c;
}

}  // namespace ojf

OJF_API const char *ojf_version(void) { return "ojf 0.1.0 (gfx950)"; }

OJF_API const char *ojf_last_error(void) { return ojf::g_last_error.c_str(); }

OJF_API int ojf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return ojf::check_hip(e, "hipGetDeviceCount");
    if (n == 0) return ojf::fail("no HIP device visible");
    return n;
}
