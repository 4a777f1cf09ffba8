// knobs.hpp -- the registry of environment knobs (knobs.cpp): the only way the engine reads its environment.
#pragma once
namespace mcgpu {
struct KnobSpec { const char* name; char type; char scope; const char* dflt; const char* what; };
extern const KnobSpec kKnobs[];
extern const int kNumKnobs;
const char* knob_str(const char* name);            // value or nullptr; throws mcgpu::Error for a name that is not registered
bool knob_set(const char* name);
int knob_int(const char* name, int dflt);
double knob_float(const char* name, double dflt);
void knobs_warn_unknown();                          // one line per unregistered MCGPU_* variable, once per process
}  // namespace mcgpu
