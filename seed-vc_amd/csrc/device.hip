// Per-device lazily created state (zero page, one-time kernel attributes).  A process may drive several devices
// (one handle per device, the device current when the handle is used); the state is keyed by hipGetDevice and created
// once per device under a mutex, so concurrent first launches from several host threads are safe.
#include <map>
#include <mutex>

#include "common.h"

namespace svc {

int current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

DeviceState* device_state() {
    static std::mutex mu;
    static std::map<int, DeviceState*> states;
    const int d = current_device();
    if (d < 0) {
        set_error("hipGetDevice failed");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    auto it = states.find(d);
    if (it != states.end()) return it->second;
    void* zp = nullptr;
    if (hipMalloc(&zp, 256) != hipSuccess || hipMemset(zp, 0, 256) != hipSuccess) {
        set_error("device_state: zero page allocation failed");
        return nullptr;
    }
    DeviceState* s = new DeviceState();
    s->device = d;
    s->zero_page = zp;
    states[d] = s;
    return s;
}

}  // namespace svc
