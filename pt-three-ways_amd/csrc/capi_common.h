// capi_common.h — shared by the two halves of the C ABI (capi_host.cpp, capi_render.hip).
#pragma once

#include "../../include/ptw.h"

#include <fstream>
#include <stdexcept>
#include <string>

namespace ptw {

// A failed HIP call or launch; carries the ptw_status to return.
struct DeviceError : std::runtime_error {
  int status;
  DeviceError(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

void setLastError(const std::string &message);
// Call inside a catch(...) block: records the message and maps the exception to a ptw_status.
int translateException();
int invalid(const char *what);

} // namespace ptw
