// graphlily/app/module_collection.h -- owner of a driver's modules (reference
// app/module_collection.h:13-114).  One device context serves all modules; there is no bitstream.
#ifndef GRAPHLILY_MODULE_COLLECTION_H_
#define GRAPHLILY_MODULE_COLLECTION_H_

#include <cassert>
#include <string>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/module/base_module.h"

namespace graphlily {
namespace app {

using namespace module;

class ModuleCollection {
protected:
    std::vector<BaseModule *> modules_;
    uint32_t num_modules_ = 0;
    std::vector<std::string> kernel_names_;
    std::string target_ = "hw";
    int device_ = 0;

public:
    ModuleCollection() {}
    ~ModuleCollection() {
        for (BaseModule *m : modules_) delete m;  // takes ownership, like the reference (:36-40)
    }

    void add_module(BaseModule *module) {
        modules_.push_back(module);
        module->set_owner(this);   // modules of one collection that hold the same matrix are paired (module/fusion.h)
        kernel_names_.push_back(module->get_kernel_name());
        num_modules_++;
    }

    void set_target(std::string target) {
        assert(target == "sw_emu" || target == "hw_emu" || target == "hw");
        target_ = target;
    }

    void set_device(int device) { device_ = device; }  // extension

    void set_up_runtime(std::string /*xclbin_file_path*/) {
        GRAPHLILY_CHECK(gl_init(device_));
        for (BaseModule *m : modules_) {
            m->set_device(device_);
            m->set_unused_args();
            m->set_mode();
            // one in-order stream serves all modules; host-visible calls synchronise (base_module.h).
            // GRAPHLILY_BLOCKING=1 restores a finish() after every launch (time-breakdown style measurements).
            m->set_blocking(getenv("GRAPHLILY_BLOCKING") != nullptr && atoi(getenv("GRAPHLILY_BLOCKING")) != 0);
        }
    }
};

}  // namespace app
}  // namespace graphlily

#endif  // GRAPHLILY_MODULE_COLLECTION_H_
