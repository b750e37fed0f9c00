// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hlsl_shim.h).
//
// Stand-in for NVIDIA-RTX/MathLib's "ml.hlsli", which every reference shader includes and which is NOT part of /root/reference (fetched unpinned at
// configure time, reference CMakeLists.txt:118-127). PARITY UNPINNED for this file: the functions below restate MathLib from its public behaviour and
// from the anchors the reference itself holds (SURVEY.md section 8c) -- the same definitions as oracle/ml.h, written in the shaders' own language and
// in plain IEEE arithmetic. Everything else oracle/_ref executes IS the reference's text.
// Goes through the same text pipeline as the shaders (hlsl2cpp.py), so it is written in the common subset of HLSL and C++.
#ifndef ML_HLSLI_STANDIN
#define ML_HLSLI_STANDIN

#define compiletime

namespace Math
{
    float Pi( float x ) { return 3.14159265358979323846 * x; }
    float DegToRad( float x ) { return x * ( 3.14159265358979323846 / 180.0 ); }
    float LinearStep( float a, float b, float x ) { return saturate( ( x - a ) / ( b - a ) ); }
    float2 LinearStep( float a, float b, float2 x ) { return saturate( ( x - a ) / ( b - a ) ); }
    float SmoothStep01( float x ) { x = saturate( x ); return x * x * ( 3.0 - 2.0 * x ); }
    float4 SmoothStep01( float4 x ) { x = saturate( x ); return x * x * ( 3.0 - 2.0 * x ); }
    float SmoothStep( float a, float b, float x ) { return SmoothStep01( LinearStep( a, b, x ) ); }
    float4 SmoothStep( float a, float b, float4 x ) { return SmoothStep01( saturate( ( x - a ) / ( b - a ) ) ); }
    float Sqrt01( float x ) { return sqrt( saturate( x ) ); }
    float2 Sqrt01( float2 x ) { return sqrt( saturate( x ) ); }
    float4 Sqrt01( float4 x ) { return sqrt( saturate( x ) ); }
    float Pow01( float x, float y ) { return pow( saturate( x ), y ); }
    float2 Pow01( float2 x, float y ) { return pow( saturate( x ), y ); }
    float PositiveRcp( float x ) { return 1.0 / max( x, 1e-15 ); }
    float AcosApprox( float x ) { return 1.41421356 * sqrt( saturate( 1.0 - x ) ); } // sqrt( 2 ) * sqrt( saturate( 1 - x ) )
    float LengthSquared( float2 v ) { return dot( v, v ); }
    float LengthSquared( float3 v ) { return dot( v, v ); }
    float Rsqrt( float x ) { return rsqrt( x ); }
    uint ReverseBits4( uint x ) { return ( ( x & 1u ) << 3 ) | ( ( x & 2u ) << 1 ) | ( ( x & 4u ) >> 1 ) | ( ( x & 8u ) >> 3 ); }
}

namespace Geometry
{
    // matrices are column-major with column vectors: p' = M * p
    float3 RotateVector( float4x4 M, float3 v ) { return mul( ( float3x3 )M, v ); }
    float3 RotateVectorInverse( float4x4 M, float3 v ) { return mul( v, ( float3x3 )M ); } // = transpose( ( float3x3 )M ) * v
    float3 RotateVector( float3x3 M, float3 v ) { return mul( M, v ); }
    float3 RotateVectorInverse( float3x3 M, float3 v ) { return mul( v, M ); }
    float3 AffineTransform( float4x4 M, float3 p ) { return mul( M, float4( p, 1.0 ) ).xyz; }
    float4 ProjectiveTransform( float4x4 M, float3 p ) { return mul( M, float4( p, 1.0 ) ); }
    float2 GetScreenUv( float4x4 worldToClip, float3 X, bool killBackprojection = true )
    {
        float4 clip = ProjectiveTransform( worldToClip, X );
        float2 uv = ( clip.xy / clip.w ) * float2( 0.5, -0.5 ) + 0.5;
        return ( killBackprojection && clip.w < 0.0 ) ? float2( 99999.0, 99999.0 ) : uv;
    }
    float3 ReconstructViewPosition( float2 uv, float4 frustum, float viewZ = 1.0, float orthoMode = 0.0 )
    {
        float3 p;
        p.xy = uv * frustum.zw + frustum.xy;
        p.xy *= viewZ * ( 1.0 - abs( orthoMode ) ) + orthoMode;
        p.z = viewZ;
        return p;
    }
    // rotator = ( cos, sin, -sin, cos )
    float4 GetRotator( float angle ) { float ca = cos( angle ); float sa = sin( angle ); return float4( ca, sa, -sa, ca ); }
    float4 CombineRotators( float4 r1, float4 r2 ) { return r1.xyxy * r2.xxzz + r1.zwzw * r2.yyww; }
    float2 RotateVector( float4 rotator, float2 v ) { return v.x * rotator.xz + v.y * rotator.yw; }
    float4 ScaleRotator( float4 r, float2 s ) { return r * s.xxyy; }
    // branch-free orthonormal basis (Duff et al. 2017); rows T, B, N
    float3x3 GetBasis( float3 N )
    {
        float sz = N.z >= 0.0 ? 1.0 : -1.0;
        float a = 1.0 / ( sz + N.z );
        float ya = N.y * a;
        float b = N.x * ya;
        float c = N.x * sz;
        float3 T = float3( c * N.x * a - 1.0, sz * b, c );
        float3 B = float3( b, N.y * ya - sz, N.y );
        return float3x3( T, B, N );
    }
}

namespace Color
{
    float Luminance( float3 c ) { return c.x * 0.2126 + c.y * 0.7152 + c.z * 0.0722; } // NRD.hlsli:350-354 "must be in sync with ML_LUMINANCE_DEFAULT"
    float3 RgbToYCoCg( float3 c ) { return float3( c.x * 0.25 + c.y * 0.5 + c.z * 0.25, c.x * 0.5 + c.y * 0.0 + c.z * -0.5, c.x * -0.25 + c.y * 0.5 + c.z * -0.25 ); } // NRD.hlsli:356-363
    float3 YCoCgToRgb( float3 c ) { float t = c.x - c.z; return float3( max( t + c.y, 0.0 ), max( c.x + c.z, 0.0 ), max( t - c.y, 0.0 ) ); } // NRD.hlsli:365-375
    float Clamp( float m1, float sigma, float x ) { return clamp( x, m1 - sigma, m1 + sigma ); }
    float2 Clamp( float2 m1, float2 sigma, float2 x ) { return clamp( x, m1 - sigma, m1 + sigma ); }
    float3 Clamp( float3 m1, float3 sigma, float3 x ) { return clamp( x, m1 - sigma, m1 + sigma ); }
    float4 Clamp( float4 m1, float4 sigma, float4 x ) { return clamp( x, m1 - sigma, m1 + sigma ); }
    float ZucconiBump( float x, float yoffset ) { return saturate( ( 1.0 - x * x ) - yoffset ); }
    float3 ColorizeZucconi( float x )
    {
        x = saturate( x );
        return float3( ZucconiBump( 3.54585104 * ( x - 0.69549072 ), 0.02312639 ) + ZucconiBump( 3.90307140 * ( x - 0.11748627 ), 0.84897130 ),
                       ZucconiBump( 2.93225262 * ( x - 0.49228336 ), 0.15225084 ) + ZucconiBump( 3.21182957 * ( x - 0.86755042 ), 0.88445281 ),
                       ZucconiBump( 2.41593945 * ( x - 0.27699880 ), 0.52607955 ) + ZucconiBump( 3.96587128 * ( x - 0.66077860 ), 0.73949448 ) );
    }
}

// MathLib's debug text (font tables, Text::Print_*): NOT available and not restated -- the overlay shaders are compiled with a Text that prints nothing, which is what the
// HIP library's validation kernels implement ("without MathLib's debug text", DESIGN.md section 1). Everything else the *_Validation shaders draw is the reference's own text.
namespace Text
{
    static const uint Char_Minus = 45;
    uint4 Init( int2 pixelPos, float2 origin, uint scale ) { return uint4( 0, 0, 0, 0 ); }
    void Print_ch( uint c, inout uint4 state ) {}
    bool IsForeground( uint4 state ) { return false; }
}

namespace Packing
{
    // round( saturate( c ) * ( 2^bits - 1 ) ), least significant field first
    uint RgbaToUint( float4 c, compiletime const uint Rbits, compiletime const uint Gbits, compiletime const uint Bbits, compiletime const uint Abits )
    {
        const uint Rmask = ( 1u << Rbits ) - 1u;
        const uint Gmask = ( 1u << Gbits ) - 1u;
        const uint Bmask = ( 1u << Bbits ) - 1u;
        const uint Amask = ( 1u << Abits ) - 1u;
        const uint Gshift = Rbits;
        const uint Bshift = Gshift + Gbits;
        const uint Ashift = Bshift + Bbits;
        const float4 scale = float4( float( Rmask ), float( Gmask ), float( Bmask ), float( Amask ) );
        uint4 p = uint4( saturate( c ) * scale + 0.5 );
        return p.x | ( p.y << Gshift ) | ( p.z << Bshift ) | ( p.w << Ashift );
    }
    float4 UintToRgba( uint p, compiletime const uint Rbits, compiletime const uint Gbits, compiletime const uint Bbits, compiletime const uint Abits )
    {
        const uint Rmask = ( 1u << Rbits ) - 1u;
        const uint Gmask = ( 1u << Gbits ) - 1u;
        const uint Bmask = ( 1u << Bbits ) - 1u;
        const uint Amask = ( 1u << Abits ) - 1u;
        const uint Gshift = Rbits;
        const uint Bshift = Gshift + Gbits;
        const uint Ashift = Bshift + Bbits;
        // field / ( 2^bits - 1 ) as a true quotient. [ml ambiguity] A reciprocal scale ( v * ( 1.0 / mask ) ), the other plausible MathLib form, is not exact: with the 4-bit
        // material field of REBLUR's internal data ( mask 15 ) the IDs 3, 6, 7, 12, 13, 14 unpack to m * 15 * ( 1 / 15 ) = m + 1 ulp and CompareMaterials( m, m ) fails --
        // material 3 of the 2-bit G-buffer field would never match its own history. The restatement ( oracle/ml.h, csrc/hip ) keeps IDs exact; so does this stand-in.
        const float4 denom = max( float4( float( Rmask ), float( Gmask ), float( Bmask ), float( Amask ) ), 1.0 );
        uint4 v = uint4( p & Rmask, ( p >> Gshift ) & Gmask, ( p >> Bshift ) & Bmask, ( p >> Ashift ) & Amask );
        return float4( v ) / denom;
    }
}

namespace Sequence
{
    uint CheckerBoard( uint2 p, uint frameIndex ) { return ( ( p.x ^ p.y ) ^ frameIndex ) & 1u; }
    uint Bayer4x4ui( uint2 p, uint frameIndex )
    {
        uint2 q = p & 3u;
        uint a = 2068378560u * ( 1u - ( q.x >> 1 ) ) + 1500172770u * ( q.x >> 1 );
        uint b = ( q.y + ( ( q.x & 1u ) << 2 ) ) << 2;
#if( NRD_MATHLIB_BAYER_REVERSEBITS == 1 ) // the alternative the round-5 review recalls as MathLib's default (ML_BAYER_REVERSEBITS); default here: 0, see oracle/ref/Makefile MATHLIB_DEFS
        uint sampleOffset = Math::ReverseBits4( frameIndex );
#else
        uint sampleOffset = frameIndex;
#endif
        return ( ( a >> b ) + sampleOffset ) & 0xFu;
    }
    float Bayer4x4( uint2 p, uint frameIndex ) { return float( Bayer4x4ui( p, frameIndex ) ) * 0.0625; } // RESULT: [0; 1) (round 5: i / 16)
}

namespace Rng
{
    // Our own hash (MathLib's is unavailable): seeded per ( pixel, frame ), PCG output function -- the definition of oracle/ml.h RngHash
    namespace Hash
    {
        static thread_local uint g_State = 0u;
        void Initialize( uint2 p, uint frameIndex )
        {
            uint s = p.x * 0x9E3779B1u ^ ( p.y * 0x85EBCA77u + 0xC2B2AE3Du ) ^ ( frameIndex * 0x27D4EB2Fu + 0x165667B1u );
            s ^= s >> 15;
            s *= 0x2C1B3C6Du;
            s ^= s >> 12;
            s *= 0x297A2D39u;
            s ^= s >> 15;
            g_State = s;
        }
        uint Next( )
        {
            g_State = g_State * 747796405u + 2891336453u;
            uint w = ( ( g_State >> ( ( g_State >> 28 ) + 4u ) ) ^ g_State ) * 277803737u;
            return ( w >> 22 ) ^ w;
        }
        float GetFloat( ) { return float( Next( ) >> 8 ) * ( 1.0 / 16777216.0 ); }
        float2 GetFloat2( ) { float a = GetFloat( ); float b = GetFloat( ); return float2( a, b ); }
    }
}

namespace Filtering
{
    struct Bilinear { float2 origin; float2 weights; };
    struct CatmullRom { float2 origin; float2 weights[ 4 ]; };

    Bilinear GetBilinearFilter( float2 uv, float2 texSize )
    {
        float2 t = uv * texSize - 0.5;
        Bilinear result;
        result.origin = floor( t );
        result.weights = t - result.origin;
        return result;
    }
    float ApplyBilinearFilter( float s00, float s10, float s01, float s11, Bilinear f ) { return lerp( lerp( s00, s10, f.weights.x ), lerp( s01, s11, f.weights.x ), f.weights.y ); }
    float2 ApplyBilinearFilter( float2 s00, float2 s10, float2 s01, float2 s11, Bilinear f ) { return lerp( lerp( s00, s10, f.weights.x ), lerp( s01, s11, f.weights.x ), f.weights.y ); }
    float3 ApplyBilinearFilter( float3 s00, float3 s10, float3 s01, float3 s11, Bilinear f ) { return lerp( lerp( s00, s10, f.weights.x ), lerp( s01, s11, f.weights.x ), f.weights.y ); }
    float4 ApplyBilinearFilter( float4 s00, float4 s10, float4 s01, float4 s11, Bilinear f ) { return lerp( lerp( s00, s10, f.weights.x ), lerp( s01, s11, f.weights.x ), f.weights.y ); }
    float4 GetBilinearCustomWeights( Bilinear f, float4 customWeights )
    {
        float2 oneMinusWeights = 1.0 - f.weights;
        float4 weights = customWeights;
        weights.x *= oneMinusWeights.x * oneMinusWeights.y;
        weights.y *= f.weights.x * oneMinusWeights.y;
        weights.z *= oneMinusWeights.x * f.weights.y;
        weights.w *= f.weights.x * f.weights.y;
        return weights;
    }
    // mirrors Common.hlsli:655-656: 0 when the weights vanish
    float ApplyBilinearCustomWeights( float s00, float s10, float s01, float s11, float4 w )
    {
        float r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
        float sumw = dot( w, 1.0 );
        return sumw < 0.0001 ? 0.0 : r / sumw;
    }
    float2 ApplyBilinearCustomWeights( float2 s00, float2 s10, float2 s01, float2 s11, float4 w )
    {
        float2 r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
        float sumw = dot( w, 1.0 );
        return sumw < 0.0001 ? float2( 0.0, 0.0 ) : r / sumw;
    }
    float3 ApplyBilinearCustomWeights( float3 s00, float3 s10, float3 s01, float3 s11, float4 w )
    {
        float3 r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
        float sumw = dot( w, 1.0 );
        return sumw < 0.0001 ? float3( 0.0, 0.0, 0.0 ) : r / sumw;
    }
    float4 ApplyBilinearCustomWeights( float4 s00, float4 s10, float4 s01, float4 s11, float4 w )
    {
        float4 r = s00 * w.x + s10 * w.y + s01 * w.z + s11 * w.w;
        float sumw = dot( w, 1.0 );
        return sumw < 0.0001 ? float4( 0.0, 0.0, 0.0, 0.0 ) : r / sumw;
    }
    // origin = top-left texel of the 4x4 footprint (REBLUR_TemporalAccumulation.hlsli:152-171 fixes the convention)
    CatmullRom GetCatmullRomFilter( float2 uv, float2 texSize, float sharpness = 0.5 )
    {
        float2 tci = uv * texSize;
        float2 tc = floor( tci - 0.5 ) + 0.5;
        float2 f = saturate( tci - tc );
        float2 f2 = f * f;
        float2 f3 = f2 * f;
        CatmullRom result;
        result.origin = tc - 1.5;
        result.weights[ 0 ] = -sharpness * f3 + 2.0 * sharpness * f2 - sharpness * f;
        result.weights[ 1 ] = ( 2.0 - sharpness ) * f3 - ( 3.0 - sharpness ) * f2 + 1.0;
        result.weights[ 2 ] = -( 2.0 - sharpness ) * f3 + ( 3.0 - 2.0 * sharpness ) * f2 + sharpness * f;
        result.weights[ 3 ] = sharpness * f3 - sharpness * f2;
        return result;
    }
    float GetModifiedRoughnessFromNormalVariance( float linearRoughness, float3 nonNormalizedAverageNormal )
    {
        float l = length( nonNormalizedAverageNormal );
        float kappa = saturate( 1.0 - l * l ) / max( l * ( 3.0 - l * l ), 1e-15 );
        return sqrt( saturate( linearRoughness * linearRoughness + kappa ) );
    }
}

namespace ImportanceSampling
{
    // "the MathLib one has been fixed" (RELAX_Common.hlsli:113-122, caller squares the fraction: Reblur.cpp:384): m * sqrt( p / ( 1 - p ) ), m = roughness^2
    float GetSpecularLobeTanHalfAngle( float linearRoughness, float percentOfVolume = 0.75 )
    {
        linearRoughness = saturate( linearRoughness );
        percentOfVolume = saturate( percentOfVolume );
        float m = linearRoughness * linearRoughness;
        return m * sqrt( percentOfVolume / ( 1.0 - percentOfVolume + 1e-6 ) );
    }
    float GetSpecularDominantFactor( float NoV, float linearRoughness, compiletime const uint mode = 0 ) // NRD.hlsli:386-392
    {
        float a = 0.298475 * log( 39.4115 - 39.0029 * linearRoughness );
        float dominantFactor = pow( saturate( 1.0 - NoV ), 10.8649 ) * ( 1.0 - a ) + a;
        return saturate( dominantFactor );
    }
    float4 GetSpecularDominantDirection( float3 N, float3 V, float linearRoughness, compiletime const uint mode = 0 ) // NRD.hlsli:394-400
    {
        float NoV = abs( dot( N, V ) );
        float dominantFactor = GetSpecularDominantFactor( NoV, linearRoughness, mode );
        float3 R = reflect( -V, N );
        float3 D = normalize( lerp( N, R, dominantFactor ) );
        return float4( D, dominantFactor );
    }
}

#define ML_SPECULAR_DOMINANT_DIRECTION_G2 0
#define ML_SPECULAR_DOMINANT_DIRECTION_G1 1
#define ML_SPECULAR_DOMINANT_DIRECTION_DEFAULT ML_SPECULAR_DOMINANT_DIRECTION_G2

namespace BRDF
{
    float Pow5( float x ) { float t = saturate( 1.0 - x ); float t2 = t * t; return t2 * t2 * t; } // NRD.hlsli:408-411
    void ConvertBaseColorMetalnessToAlbedoRf0( float3 baseColor, float metalness, out float3 albedo, out float3 Rf0 )
    {
        albedo = baseColor * saturate( 1.0 - metalness );
        Rf0 = 0.04 + ( baseColor - 0.04 ) * metalness;
    }
    float3 EnvironmentTerm_Rtg( float3 Rf0, float NoV, float linearRoughness ) // NRD.hlsli:490-517
    {
        float m = saturate( linearRoughness * linearRoughness );
        float4 X;
        X.x = 1.0;
        X.y = NoV;
        X.z = NoV * NoV;
        X.w = NoV * X.z;
        float4 Y;
        Y.x = 1.0;
        Y.y = m;
        Y.z = m * m;
        Y.w = m * Y.z;
        float2x2 M1 = float2x2( 0.99044, -1.28514, 1.29678, -0.755907 );
        float3x3 M2 = float3x3( 1.0, 2.92338, 59.4188, 20.3225, -27.0302, 222.592, 121.563, 626.13, 316.627 );
        float2x2 M3 = float2x2( 0.0365463, 3.32707, 9.0632, -9.04756 );
        float3x3 M4 = float3x3( 1.0, 3.59685, -1.36772, 9.04401, -16.3174, 9.22949, 5.56589, 19.7886, -20.2123 );
        float bias = dot( mul( M1, X.xy ), Y.xy ) * rcp( max( dot( mul( M2, X.xyw ), Y.xyw ), 1e-6 ) );
        float scale = dot( mul( M3, X.xy ), Y.xy ) * rcp( max( dot( mul( M4, X.xzw ), Y.xyw ), 1e-6 ) );
        return saturate( Rf0 * scale + bias );
    }
}

#endif
