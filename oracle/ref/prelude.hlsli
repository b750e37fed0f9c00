// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hlsl_shim.h).
// Force-included in front of every reference shader entry by hlsl2cpp.py (preprocessor stage: `clang -E -x c -undef`). It plays the role of the
// "custom engine that defined all the macros" that reference Shaders/Include/NRD.hlsli:104-118 provides for: bindings are declared through
// macros, and here they expand to HLSL_CONSTANT / HLSL_INPUT / HLSL_OUTPUT, which oracle/ref/hlsl_shim.h turns into C++ globals + registrations.
#define NRD_CONSTANTS_START( resourceName )
#define NRD_CONSTANT( constantType, constantName )                          HLSL_CONSTANT( constantType, constantName )
#define NRD_CONSTANTS_END
#define NRD_INPUTS_START
#define NRD_INPUT( resourceType, resourceName, regName, bindingIndex )      HLSL_INPUT( resourceType, resourceName, bindingIndex )
#define NRD_INPUTS_END
#define NRD_OUTPUTS_START
#define NRD_OUTPUT( resourceType, resourceName, regName, bindingIndex )     HLSL_OUTPUT( resourceType, resourceName, bindingIndex )
#define NRD_OUTPUTS_END
#define NRD_SAMPLERS_START
#define NRD_SAMPLER( resourceType, resourceName, regName, bindingIndex )
#define NRD_SAMPLERS_END
#define NRD_CS_MAIN                                                         hlsl_cs_main
#define NRD_INTERNAL
