// REBLUR pass registry: concatenates the tables of the three REBLUR translation units.
#include "passes.h"

#include <vector>

namespace nrdhip {

const PassEntry* GetReblurSpatialPasses(uint32_t& num);
const PassEntry* GetReblurTemporalAccumulationPasses(uint32_t& num);
const PassEntry* GetReblurHistoryPasses(uint32_t& num);

const PassEntry* GetReblurPasses(uint32_t& num) {
    static std::vector<PassEntry> all = [] {
        std::vector<PassEntry> v;
        uint32_t n = 0;
        const PassEntry* t = GetReblurSpatialPasses(n);
        v.insert(v.end(), t, t + n);
        t = GetReblurTemporalAccumulationPasses(n);
        v.insert(v.end(), t, t + n);
        t = GetReblurHistoryPasses(n);
        v.insert(v.end(), t, t + n);
        return v;
    }();
    num = (uint32_t)all.size();
    return all.data();
}

} // namespace nrdhip
