// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE, and not a file of the reference: a compute shader written for this repository that does nothing but CALL the reference's
// application-side functions (Shaders/Include/NRD.hlsli: front-end packers, back-end unpackers, material factors, SG / SH resolves, re-jitter) on rows of inputs, so that
// include/NRD.hip.h -- the product's counterpart of that header -- is held against the reference's own text (tests/test_frontend_header.py). Compiled like every reference entry
// (oracle/ref/hlsl2cpp.py -> oracle/_ref/libnrdref.so); NRD.hlsli is read where it lies. The call sequence mirrors tests/cpp/frontend_check.hip Evaluate().
#include "NRD.hlsli"

NRD_CONSTANTS_START( NRD_FrontEndProbeConstants )
    NRD_CONSTANT( float4, gHitDistParams )
    NRD_CONSTANT( uint, gWidth )
NRD_CONSTANTS_END

NRD_INPUTS_START
    NRD_INPUT( Texture2D<float4>, gIn_N_Roughness, t, 0 )
    NRD_INPUT( Texture2D<float4>, gIn_V_MaterialID, t, 1 )
    NRD_INPUT( Texture2D<float4>, gIn_Radiance_HitDist, t, 2 )
    NRD_INPUT( Texture2D<float4>, gIn_Direction_ViewZ, t, 3 )
    NRD_INPUT( Texture2D<float4>, gIn_Albedo_Miss, t, 4 )
    NRD_INPUT( Texture2D<float4>, gIn_Rf0, t, 5 )
    NRD_INPUT( Texture2D<float4>, gIn_Nw, t, 6 )
    NRD_INPUT( Texture2D<float4>, gIn_PackedNormalRoughness, t, 7 )
NRD_INPUTS_END

NRD_OUTPUTS_START
    NRD_OUTPUT( RWTexture2D<float4>, gOut_PackedNormalRoughness, u, 0 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_UnpackedNR, u, 1 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_Scalars, u, 2 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_ReblurPacked, u, 3 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_ReblurUnpacked, u, 4 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_Sh0, u, 5 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_Sh1, u, 6 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_RelaxPacked, u, 7 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_RelaxSh1, u, 8 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_DirOcc, u, 9 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_Translucency, u, 10 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_DiffFactor_MaterialID, u, 11 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_SpecFactor, u, 12 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_SgDiffuse, u, 13 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_SgSpecular, u, 14 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_ShDiffuse, u, 15 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_ShSpecular, u, 16 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_SgColor, u, 17 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_SgDir, u, 18 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_ReJitter, u, 19 )
    NRD_OUTPUT( RWTexture2D<float4>, gOut_Misc, u, 20 )
NRD_OUTPUTS_END

[numthreads( 8, 8, 1 )]
NRD_EXPORT void NRD_CS_MAIN( uint2 pixelPos : SV_DispatchThreadId )
{
    if( pixelPos.x >= gWidth )
        return;

    float4 a = gIn_N_Roughness[ pixelPos ];
    float4 b = gIn_V_MaterialID[ pixelPos ];
    float4 c = gIn_Radiance_HitDist[ pixelPos ];
    float4 d = gIn_Direction_ViewZ[ pixelPos ];
    float4 e = gIn_Albedo_Miss[ pixelPos ];
    float3 N = a.xyz;
    float roughness = a.w;
    float3 V = b.xyz;
    float3 radiance = c.xyz;
    float hitDist = c.w;
    float3 direction = d.xyz;
    float viewZ = d.w;
    float3 albedo = e.xyz;
    float3 Rf0 = gIn_Rf0[ pixelPos ].xyz;
    float3 Nw = gIn_Nw[ pixelPos ].xyz;
    float occluder = e.w != 0.0 ? NRD_FP16_MAX : hitDist;

    gOut_PackedNormalRoughness[ pixelPos ] = NRD_FrontEnd_PackNormalAndRoughness( N, roughness, b.w );

    float materialID;
    gOut_UnpackedNR[ pixelPos ] = NRD_FrontEnd_UnpackNormalAndRoughness( gIn_PackedNormalRoughness[ pixelPos ], materialID );

    float normHitDist = REBLUR_FrontEnd_GetNormHitDist( hitDist, viewZ, gHitDistParams, roughness );
    gOut_Scalars[ pixelPos ] = float4( normHitDist, SIGMA_FrontEnd_PackPenumbra( occluder, 0.02 ), SIGMA_FrontEnd_PackPenumbra( hitDist, hitDist + 10.0, 0.5 ), SIGMA_BackEnd_UnpackShadow( roughness ) );

    float4 reblurPacked = REBLUR_FrontEnd_PackRadianceAndNormHitDist( radiance, normHitDist, true );
    gOut_ReblurPacked[ pixelPos ] = reblurPacked;
    gOut_ReblurUnpacked[ pixelPos ] = REBLUR_BackEnd_UnpackRadianceAndNormHitDist( reblurPacked );

    float4 sh1;
    float4 sh0 = REBLUR_FrontEnd_PackSh( radiance, normHitDist, direction, sh1, true );
    gOut_Sh0[ pixelPos ] = sh0;
    gOut_Sh1[ pixelPos ] = sh1;

    float4 relaxSh1;
    gOut_RelaxPacked[ pixelPos ] = RELAX_FrontEnd_PackSh( radiance, hitDist, direction, relaxSh1, true );
    gOut_RelaxSh1[ pixelPos ] = relaxSh1;

    gOut_DirOcc[ pixelPos ] = REBLUR_FrontEnd_PackDirectionalOcclusion( direction, normHitDist, true );
    gOut_Translucency[ pixelPos ] = SIGMA_FrontEnd_PackTranslucency( occluder, albedo );

    float3 diffFactor, specFactor;
    NRD_MaterialFactors( N, V, albedo, Rf0, roughness, diffFactor, specFactor );
    gOut_DiffFactor_MaterialID[ pixelPos ] = float4( diffFactor, materialID );
    gOut_SpecFactor[ pixelPos ] = float4( specFactor, 0.0 );

    NRD_SG sg = REBLUR_BackEnd_UnpackSh( sh0, sh1 );
    gOut_SgColor[ pixelPos ] = float4( NRD_SG_ExtractColor( sg ), 0.0 );
    gOut_SgDir[ pixelPos ] = float4( NRD_SG_ExtractDirection( sg ), 0.0 );
    gOut_SgDiffuse[ pixelPos ] = float4( NRD_SG_ResolveDiffuse( sg, N ), 0.0 );
    gOut_SgSpecular[ pixelPos ] = float4( NRD_SG_ResolveSpecular( sg, N, V, roughness ), 0.0 );
    gOut_ShDiffuse[ pixelPos ] = float4( NRD_SH_ResolveDiffuse( sg, N ), 0.0 );
    gOut_ShSpecular[ pixelPos ] = float4( NRD_SH_ResolveSpecular( sg, N, V, roughness ), 0.0 );
    gOut_ReJitter[ pixelPos ] = float4( NRD_SG_ReJitter( sg, sg, Rf0, V, roughness, viewZ, viewZ * 1.001, viewZ * 0.999, viewZ, viewZ, N, N, Nw, N, N ), 0.0, 0.0 );

    // MISC ( NRD.hlsli:575-580, 1136-1162 )
    NRD_SG wide = sg;
    wide.sharpness = 0.5 + 4.0 * roughness;
    float poison = e.w != 0.0 ? 0.0 : 1.0;
    gOut_Misc[ pixelPos ] = float4( REBLUR_GetHitDist( normHitDist, viewZ, gHitDistParams, roughness ), NRD_GetNormalizedStrandThickness( hitDist * 0.01, viewZ * 0.001 ), _NRD_SG_Integral( wide ),
        NRD_IsValidRadiance( radiance / poison ) ? 1.0 : 0.0 );
}
