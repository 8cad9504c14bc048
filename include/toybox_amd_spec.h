/*
 * toybox_amd_spec.h -- numeric constants and sprite bitmaps of this repo's game specifications
 * (SPEC.md).  Data only: the HIP kernels and the CPU checker each state
 * the rules themselves and share just these tables.  Values anchored on the reference's golden
 * dumps are marked [golden]; the rest is this repo's own choice (the reference's Rust core,
 * ctoybox==0.5.0, is not in the reference tree).
 */
#ifndef TOYBOX_AMD_SPEC_H
#define TOYBOX_AMD_SPEC_H

#include <stdint.h>

/* 3x5 digit font shared by every HUD: bit (row*3+col), col 0 = left; drawn 2x (6x10 px) */
#define TBX_DIGIT_FONT {0x7B6F, 0x749A, 0x73E7, 0x79E7, 0x49ED, 0x79CF, 0x7BCF, 0x4927, 0x7BEF, 0x79EF}

/* ---- SpaceInvaders ---- */
#define TBX_SI_ENEMY_X0        44    /* [golden] enemies at (44+32c, 31+18r), 16x10 */
#define TBX_SI_ENEMY_Y0        31
#define TBX_SI_ENEMY_DX        32
#define TBX_SI_ENEMY_DY        18
#define TBX_SI_ENEMY_W         16
#define TBX_SI_ENEMY_H         10
#define TBX_SI_SHIP_X0         68    /* [golden] ship (68,185) 16x10 speed 3 */
#define TBX_SI_SHIP_Y          185
#define TBX_SI_SHIP_W          16
#define TBX_SI_SHIP_H          10
#define TBX_SI_SHIP_SPEED      3
#define TBX_SI_SHIP_X_MIN      38
#define TBX_SI_SHIP_X_MAX      266
#define TBX_SI_UFO_X0          (-2)  /* [golden] ufo (-2,12), appearance_counter 500 */
#define TBX_SI_UFO_Y           12
#define TBX_SI_UFO_W           21
#define TBX_SI_UFO_H           13
#define TBX_SI_UFO_PERIOD      500
#define TBX_SI_UFO_STEP        2
#define TBX_SI_UFO_BONUS       100
#define TBX_SI_NEW_LIFE_TIME   128   /* [golden] life_display_timer */
#define TBX_SI_SHOT_DELAY      50    /* [golden] enemy_shot_delay */
#define TBX_SI_MOVE_PERIOD     32    /* [golden] per-enemy move_counter of the old dump */
#define TBX_SI_MOVE_PERIOD_MIN 4
#define TBX_SI_STEP_X          2
#define TBX_SI_STEP_Y          10
#define TBX_SI_FIELD_X_MIN     22
#define TBX_SI_FIELD_X_MAX     298
#define TBX_SI_GROUND_Y        195
#define TBX_SI_ENEMY_DEATH_T   16
#define TBX_SI_SHIP_DEATH_T    32
#define TBX_SI_UFO_DEATH_T     32
#define TBX_SI_LASER_W         2
#define TBX_SI_LASER_H         8
#define TBX_SI_SHIP_LASER_V    6
#define TBX_SI_ENEMY_LASER_V   3

/* sprites: one uint32 per row, bit c = pixel column c (col 0 = left) */
#define TBX_SI_SPRITE_ENEMY_A {0x0420, 0x0240, 0x07E0, 0x0DB0, 0x1FF8, 0x17E8, 0x1428, 0x0360, 0x0000, 0x0000}
#define TBX_SI_SPRITE_ENEMY_B {0x0420, 0x1248, 0x17E8, 0x1DB8, 0x1FF8, 0x0FF0, 0x0420, 0x0810, 0x0000, 0x0000}
#define TBX_SI_SPRITE_BOOM    {0x0000, 0x0890, 0x0420, 0x0000, 0x1818, 0x0000, 0x0420, 0x0890, 0x0000, 0x0000}
#define TBX_SI_SPRITE_SHIP    {0x0080, 0x01C0, 0x01C0, 0x0FF8, 0x1FFC, 0x1FFC, 0x1FFC, 0x1FFC, 0x0000, 0x0000}
#define TBX_SI_SPRITE_SHIP_D1 {0x0000, 0x0220, 0x0088, 0x0A50, 0x0180, 0x1BD8, 0x0FF0, 0x1FFC, 0x0000, 0x0000}
#define TBX_SI_SPRITE_SHIP_D2 {0x0410, 0x0004, 0x1240, 0x0028, 0x0500, 0x0A90, 0x17E8, 0x0FF8, 0x0000, 0x0000}
#define TBX_SI_SPRITE_UFO     {0x007E00, 0x01FF80, 0x03FFC0, 0x06DB60, 0x0FFFF0, 0x039CE0, 0x010840, \
                               0x000000, 0x000000, 0x000000, 0x000000, 0x000000, 0x000000}

/* colours (r,g,b) */
#define TBX_SI_COL_ENEMY      134, 134, 29
#define TBX_SI_COL_UFO        151, 25, 122
#define TBX_SI_COL_SHIP       35, 129, 59      /* [golden] ship.color */
#define TBX_SI_COL_SHIELD     172, 80, 48      /* [golden] shield pixel colour */
#define TBX_SI_COL_SHIP_LASER 142, 142, 142
#define TBX_SI_COL_ENEMY_LASER 255, 255, 255
#define TBX_SI_COL_GROUND     80, 89, 22
#define TBX_SI_COL_HUD        50, 132, 50

/* ---- Amidar ---- */
#define TBX_AMI_BOARD_OX       16    /* board origin on screen; one tile = 4 x 5 px = 64 x 80 world units (scale 16) */
#define TBX_AMI_BOARD_OY       37
#define TBX_AMI_TILE_PW        4
#define TBX_AMI_TILE_PH        5
#define TBX_AMI_WORLD_SCALE    16
#define TBX_AMI_MOVER_W        6
#define TBX_AMI_MOVER_H        7
#define TBX_AMI_SPEED          8     /* [golden] speed of every mover */
#define TBX_AMI_HIT_DX         48    /* collision box, world units */
#define TBX_AMI_HIT_DY         60
#define TBX_AMI_HUD_Y          204
#define TBX_AMI_N_ROUTES       5
#define TBX_AMI_ROUTE_LEN      5
/* default EnemyLookupAI routes: tile ids ty*32+tx, -1 terminated; entry 0 is the start tile, which matches the
   golden enemy positions (0,0) (0,0) (448,0) (0,2000) (576,2400) */
#define TBX_AMI_ROUTES { {0, 31, 991, 960, -1}, {0, 192, 223, 31, -1}, {7, 10, 202, 198, 6}, \
                         {800, 768, 774, 966, 960}, {969, 972, 780, 774, 966} }

#endif
