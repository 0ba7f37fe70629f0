// A function attribute (hipFuncSetAttribute: the dynamic-LDS limit of a kernel) is a property of the function ON a device, and a
// process may drive more than one device: run the setter once per device, under a lock (a second thread's first launch on that
// device must not overtake it).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

namespace ttt {
struct OncePerDevice {
    std::mutex m;
    unsigned long long seen = 0;
    template <class F>
    void run(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        std::lock_guard<std::mutex> lock(m);
        if (!(seen & bit)) {
            f();
            seen |= bit;
        }
    }
};
}  // namespace ttt
