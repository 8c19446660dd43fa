/* FNV-1a 64 of the Layer I / II spreading function s[63][63] (float) per sampling-rate index, as the environment the
 * goldens come from computes it (printed by `make blob`; csrc/tables_host.cpp checks the blob's entries against them). */
#define L12_SPREAD_PIN_0 0xda58559f3062cbceull
#define L12_SPREAD_PIN_1 0x3f35a85f5a375bdeull
#define L12_SPREAD_PIN_2 0x0a7991f880163333ull
