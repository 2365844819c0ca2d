// host_pure.h — the thread-local detail string behind pgenhip_last_error_detail, shared by host_pure.cpp and capi.hip.
#pragma once

namespace pgenhip {
int set_detail(int status, const char *what);  // remembers `what` for this thread, returns `status`
const char *g_detail_c_str();
}  // namespace pgenhip
