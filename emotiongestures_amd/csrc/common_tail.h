// Last line of every .hip source: closes the `#pragma clang attribute push` of common.h (no-packed-fp32-ops on the device pass).
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
