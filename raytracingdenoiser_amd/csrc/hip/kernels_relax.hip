// RELAX pass registry: concatenates the tables of the three RELAX translation units.
#include "passes.h"

#include <vector>

namespace nrdhip {

const PassEntry* GetRelaxSpatialPasses(uint32_t& num);
const PassEntry* GetRelaxTemporalPasses(uint32_t& num);
const PassEntry* GetRelaxAtrousPasses(uint32_t& num);

const PassEntry* GetRelaxPasses(uint32_t& num) {
    static std::vector<PassEntry> all = [] {
        std::vector<PassEntry> v;
        uint32_t n = 0;
        const PassEntry* t = GetRelaxSpatialPasses(n);
        v.insert(v.end(), t, t + n);
        t = GetRelaxTemporalPasses(n);
        v.insert(v.end(), t, t + n);
        t = GetRelaxAtrousPasses(n);
        v.insert(v.end(), t, t + n);
        return v;
    }();
    num = (uint32_t)all.size();
    return all.data();
}

} // namespace nrdhip
