/* mem_image.c -- the "-v <mems_file>" tool of the slaMEM command line: a picture of the MEMs of every query against the one
 * reference, written as an 8-bit run-length-coded BMP next to the MEMs file (host C only, no GPU).
 *
 * Written from the reference's observable behaviour: the same picture and the same file bytes as
 *   CreateMemMapImage                 slamem.c:354-452     (what is read, what is printed, which errors end the run)
 *   the picture's geometry            graphics.c:246-379   (1024 px wide, a ruler, a two-strand colour bar for the reference,
 *                                                           one grey track per query, names in a 5x6 font)
 *   palette, nearest colour, file     bitmap.c:107-126, 250-322, 334-427, 495-633
 * Structure is this file's own: one canvas object, a layout computed once, blocks painted through a per-column "longest MEM
 * so far" table, and a byte-run encoder written as a small automaton (emit_* helpers) whose decisions are the reference's,
 * including its quirks (documented at rle8_encode), because the file is compared byte for byte (tests/test_image_tool.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slamem_host.h"

enum { GLYPH_W = 5, GLYPH_H = 6, GLYPH_GAP = 2, GLYPH_STEP = GLYPH_W + GLYPH_GAP, NUM_GLYPHS = 96, NUM_COLOURS = 255 };
enum { PIC_W = 1024, SIDE = 2 * GLYPH_W, TOP = 2 * GLYPH_H, TRACK_H = 5 * GLYPH_H, TRACK_GAP = 2 * GLYPH_H,
       RULER_H = 2 * GLYPH_H + 2 * GLYPH_GAP };

/* The font of graphics.c:26-123 (ASCII 32..126 and a filled box for everything else), one base-32 digit per glyph row, top row
   first; bit 0 of a digit is the row's leftmost pixel. */
static const char kGlyphs[NUM_GLYPHS][GLYPH_H + 1] = {
    "000000", "444404", "AA0000", "0AVAVA", "4U5EKF", "H8442H", "252L9M", "440000",
    "C2222C", "688886", "0LEVEL", "044V44", "000042", "00V000", "000004", "G84421",
    "EPLLJE", "46544V", "FGGE1V", "VGEGGF", "8CAV88", "V1FGGF", "U1FHHE", "VG8421",
    "EHEHHE", "EHHUGF", "040040", "004042", "0O616O", "00V0V0", "03CGC3", "EHC404",
    "EHTLT2", "4AEAHH", "FHFHHF", "U1111U", "FHHHHF", "V1711V", "V17111", "U11PHU",
    "HHVHHH", "V4444V", "U88996", "P5359H", "11111V", "HRLHHH", "HJLLPH", "EHHHHE",
    "FHHF11", "EHHL9M", "FHHF9H", "U1EGGF", "V44444", "HHHHHE", "HHHAA4", "HHHLRH",
    "HA4AHH", "HHA444", "V8442V", "E2222E", "12448G", "E8888E", "4AH000", "00000V",
    "240000", "0E8EAE", "22EAAE", "0E222E", "88EAAE", "0EAE2E", "C4E444", "0EAE8E",
    "22EAAA", "404444", "404446", "22A6AA", "444444", "0BLLLL", "06AAAA", "0EAAAE",
    "0EAE22", "0EAE88", "02EA22", "0E2E8E", "4E4446", "0AAAAE", "0AAA44", "0LLLAA",
    "0AA4AA", "0AAE8E", "0E842E", "C2322C", "444444", "68O886", "00IL90", "VVVVVV",
};

typedef struct {
    int w, h;
    uint8_t *px;                    /* h rows of w bytes, the picture's BOTTOM row first (the order of the file) */
    uint8_t rgb[NUM_COLOURS][3];    /* palette: white, black, grey 222, then six ramps of 42 steps round the colour circle */
} canvas;

/* ---- palette (bitmap.c:250-322) and nearest colour (bitmap.c:107-126) ---------------------------------------------------- */

static void palette_init(canvas *cv) {
    static const uint8_t fixed[3] = {255, 0, 222};
    /* each ramp moves ONE channel up (+1) or down (-1) in 42 steps of 6 (the top step is 255) while the other two stay put */
    static const int ramp[6][3] = {{255, +1, 0}, {-1, 255, 0}, {0, 255, +1}, {0, -1, 255}, {+1, 0, 255}, {255, 0, -1}};
    int at = 0, k, i, c;
    for (k = 0; k < 3; k++, at++) cv->rgb[at][0] = cv->rgb[at][1] = cv->rgb[at][2] = fixed[k];
    for (k = 0; k < 6; k++)
        for (i = 0; i < 42; i++, at++)
            for (c = 0; c < 3; c++) {
                int up = i == 41 ? 255 : 6 * i, down = i == 0 ? 255 : 6 * (41 - i);
                cv->rgb[at][c] = (uint8_t)(ramp[k][c] == +1 ? up : ramp[k][c] == -1 ? down : ramp[k][c]);
            }
}

static uint8_t colour_of(const canvas *cv, int r, int g, int b) {
    int best = 0, best_d = 3 * 255, i; /* first colour at the smallest |dr|+|dg|+|db| */
    for (i = 0; i < NUM_COLOURS; i++) {
        int d = abs(cv->rgb[i][0] - r) + abs(cv->rgb[i][1] - g) + abs(cv->rgb[i][2] - b);
        if (d == 0) return (uint8_t)i;
        if (d < best_d) { best_d = d; best = i; }
    }
    return (uint8_t)best;
}

/* ---- drawing (clipped to the canvas; y grows downwards) ------------------------------------------------------------------- */

static void dot(canvas *cv, int x, int y, uint8_t c) {
    if (x >= 0 && y >= 0 && x < cv->w && y < cv->h) cv->px[(size_t)(cv->h - 1 - y) * (size_t)cv->w + (size_t)x] = c;
}
static void across(canvas *cv, int x, int y, int len, uint8_t c) { for (; len > 0; len--) dot(cv, x++, y, c); }
static void down(canvas *cv, int x, int y, int len, uint8_t c) { for (; len > 0; len--) dot(cv, x, y++, c); }
static void box(canvas *cv, int x, int y, int w, int h, uint8_t c) { for (; h > 0; h--) across(cv, x, y++, w, c); }
static void frame(canvas *cv, int x, int y, int w, int h, uint8_t c) {
    across(cv, x, y, w, c); across(cv, x, y + h - 1, w, c);
    down(cv, x, y, h, c); down(cv, x + w - 1, y, h, c);
}

static int base32(char d) { return d <= '9' ? d - '0' : d - 'A' + 10; }

static void glyph(canvas *cv, char ch, int x, int y, uint8_t c) {
    int g = (ch >= 32 && ch <= 126) ? ch - 32 : NUM_GLYPHS - 1, r, k;
    for (r = 0; r < GLYPH_H; r++)
        for (k = 0; k < GLYPH_W; k++)
            if ((base32(kGlyphs[g][r]) >> k) & 1) dot(cv, x + k, y + r, c);
}

static void text(canvas *cv, const char *s, int max_chars, int x, int y, uint8_t c) {
    for (; *s && max_chars > 0; s++, max_chars--, x += GLYPH_STEP) glyph(cv, *s, x, y, c);
}

static int digits_of(int v) { int n = 1; while (v >= 10) { v /= 10; n++; } return n; }
static int text_px(int chars) { return chars * GLYPH_W + (chars - 1) * GLYPH_GAP; } /* C division below rounds towards 0 */

/* a number centred on x (graphics.c:188-208): never left of pixel 1 */
static void number(canvas *cv, int v, int x, int y, uint8_t c) {
    char buf[16];
    int half = text_px(digits_of(v)) / 2;
    x = x > half ? x - half : 1;
    snprintf(buf, sizeof buf, "%d", v);
    text(cv, buf, 16, x, y, c);
}

/* the bar's pointed end (graphics.c:229-244): white wedges eat the corners of a bar `h` pixels high, three columns per row */
static void wedge(canvas *cv, int x, int y, int h, int to_the_right, uint8_t c) {
    int len = ((h - 1) / 2) * 3;
    if (to_the_right) x = x - len + 1;
    for (; len > 0 && h > 0; len -= 3, y++, h -= 2) {
        across(cv, x, y, len, c);
        across(cv, x, y + h - 1, len, c);
        if (to_the_right) x += 3;
    }
}

/* ---- the picture ---------------------------------------------------------------------------------------------------------- */

typedef struct {
    canvas cv;
    int num;             /* tracks: the reference (0) and the queries                    */
    double per_pixel;    /* text positions per pixel                                      */
    int *track_w, *track_y;
    int **longest;       /* per query track and column: length of the longest MEM drawn   */
    uint8_t *strand_colour[2]; /* colour of every column of the reference bar, per strand */
    uint8_t white, black, grey;
} picture;

static int picture_init(picture *p, const int *sizes, int num) {
    int i, longest_seq = sizes[0], digits, ruler_w, step, count, room, last;
    memset(p, 0, sizeof *p);
    p->num = num;
    p->cv.w = PIC_W;
    p->cv.h = 2 * TOP + RULER_H + num * TRACK_H + (num - 1) * TRACK_GAP;
    p->cv.px = (uint8_t *)calloc((size_t)p->cv.w * (size_t)p->cv.h, 1); /* colour 0 is white */
    p->track_w = (int *)calloc((size_t)num, sizeof(int));
    p->track_y = (int *)calloc((size_t)num, sizeof(int));
    p->longest = (int **)calloc((size_t)num, sizeof(int *));
    if (!p->cv.px || !p->track_w || !p->track_y || !p->longest) return -1;
    palette_init(&p->cv);
    p->white = colour_of(&p->cv, 255, 255, 255);
    p->black = colour_of(&p->cv, 0, 0, 0);
    p->grey = colour_of(&p->cv, 222, 222, 222);
    for (i = 1; i < num; i++) if (sizes[i] > longest_seq) longest_seq = sizes[i];
    digits = digits_of(longest_seq);
    p->per_pixel = (double)longest_seq / (double)(PIC_W - 2 * SIDE - (digits * GLYPH_STEP / 2));
    for (i = 0; i < num; i++) {
        p->track_w[i] = (int)ceil((double)sizes[i] / p->per_pixel);
        p->track_y[i] = TOP + RULER_H + i * (TRACK_H + TRACK_GAP);
        if (i > 0) {
            p->longest[i] = (int *)calloc((size_t)(p->track_w[i] > 0 ? p->track_w[i] : 1), sizeof(int));
            if (!p->longest[i]) return -1;
            box(&p->cv, SIDE, p->track_y[i], p->track_w[i], TRACK_H, p->grey);
            frame(&p->cv, SIDE - 1, p->track_y[i] - 1, p->track_w[i] + 2, TRACK_H + 2, p->black);
        }
    }
    /* the reference bar: forward strand above in warm colours (magenta, red, orange, yellow), reverse strand below in cold
       ones (green, cyan, blue, purple), both running with the position (graphics.c:278-301) */
    {
        int w0 = p->track_w[0], half = TRACK_H / 2 - 1, y0 = p->track_y[0];
        double frac = 1.0 / (double)w0;
        for (i = 0; i < 2; i++)
            if (!(p->strand_colour[i] = (uint8_t *)malloc((size_t)(w0 > 0 ? w0 : 1)))) return -1;
        for (i = 0; i < w0; i++) {
            int t = (int)floor((128 + 256) * (i * frac));
            p->strand_colour[0][i] = t < 128 ? colour_of(&p->cv, 255, 0, 128 - t) : colour_of(&p->cv, 255, t - 128, 0);
            t = (int)floor((2 * 256 + 128) * (i * frac));
            p->strand_colour[1][i] = t < 256 ? colour_of(&p->cv, 0, 255, t)
                                   : t < 512 ? colour_of(&p->cv, 0, 255 - (t - 256), 255)
                                             : colour_of(&p->cv, t - 512, 0, 255);
            down(&p->cv, SIDE + i, y0, half, p->strand_colour[0][i]);
            down(&p->cv, SIDE + i, y0 + TRACK_H / 2 + 1, half, p->strand_colour[1][i]);
        }
        wedge(&p->cv, SIDE + w0 - 1, y0, half, 1, p->white);
        wedge(&p->cv, SIDE, y0 + TRACK_H / 2 + 1, half, 0, p->white);
    }
    /* the ruler (graphics.c:304-329): 1 and the longest size at the ends, between them marks at the smallest of
       1, 2, 5, 10, 20, 50, ... positions whose numbers do not touch */
    ruler_w = (int)ceil((double)longest_seq / p->per_pixel);
    number(&p->cv, 1, SIDE, TOP, p->black);
    down(&p->cv, SIDE, TOP + GLYPH_H + GLYPH_GAP, GLYPH_H, p->black);
    number(&p->cv, longest_seq, SIDE + ruler_w - 1, TOP, p->black);
    down(&p->cv, SIDE + ruler_w - 1, TOP + GLYPH_H + GLYPH_GAP, GLYPH_H, p->black);
    room = ruler_w - text_px(digits);
    count = room / (text_px(digits) + GLYPH_W);
    room = longest_seq / (count + 1);
    for (step = 1; step < room;) {
        step *= 2;         if (step >= room) break;
        step = step / 2 * 5; if (step >= room) break;
        step *= 2;
    }
    last = SIDE + ruler_w - digits * GLYPH_STEP;
    for (i = step; (count = (int)floor((double)i / p->per_pixel)) < last; i += step) {
        number(&p->cv, i, SIDE + count, TOP, p->black);
        down(&p->cv, SIDE + count, TOP + GLYPH_H + GLYPH_GAP, GLYPH_H, p->black);
    }
    across(&p->cv, SIDE, TOP + GLYPH_H + GLYPH_GAP + GLYPH_H / 2, ruler_w, p->black);
    return 0;
}

/* one MEM on a query's track (graphics.c:333-349): the columns it covers take the colour of the reference columns it matches,
   unless a longer MEM is already there.  Positions are 0-based and on the forward strand of the query. */
static void picture_block(picture *p, int track, int query_pos, int ref_pos, int len, int reverse) {
    int from = (int)floor((double)query_pos / p->per_pixel), to = (int)floor((double)(query_pos + len - 1) / p->per_pixel);
    int col = (int)floor((double)ref_pos / p->per_pixel), x;
    for (x = from; x <= to; x++, col++) {
        /* a query longer than the reference runs past the bar's last column: the reference reads behind its colour table
           there, which is untouched heap (0, white) in every run observed; this tool says white */
        uint8_t c = col < 0 || col >= p->track_w[0] ? p->white : p->strand_colour[reverse][col];
        if (x < 0 || x >= p->track_w[track]) continue;
        if (p->longest[track][x] < len) {
            down(&p->cv, SIDE + x, p->track_y[track], TRACK_H, c);
            p->longest[track][x] = len;
        }
    }
}

/* names over the tracks (graphics.c:351-366): centred on the reference bar, at the left end of a query track, as many
   characters as fit, on a white plate */
static void picture_names(picture *p, const slh_record *seqs) {
    int i;
    for (i = 0; i < p->num; i++) {
        int n = (int)strlen(seqs[i].name), fit = (p->track_w[i] - 2 * GLYPH_W) / GLYPH_STEP;
        int x, y = p->track_y[i] + TRACK_H / 2 - GLYPH_H / 2;
        if (fit < 0) fit = 0;
        if (n > fit) n = fit;
        x = i == 0 ? SIDE + p->track_w[0] / 2 - text_px(n) / 2 : SIDE + GLYPH_W;
        box(&p->cv, x - 1, y - 1, n * GLYPH_STEP - GLYPH_GAP + 2, GLYPH_H + 2, p->white);
        text(&p->cv, seqs[i].name, n, x, y, p->black);
    }
}

static void picture_free(picture *p) {
    int i;
    if (p->longest) for (i = 0; i < p->num; i++) free(p->longest[i]);
    free(p->longest); free(p->track_w); free(p->track_y);
    free(p->strand_colour[0]); free(p->strand_colour[1]);
    free(p->cv.px);
}

/* ---- the file (bitmap.c:399-427, 495-633) ---------------------------------------------------------------------------------- */

typedef struct { uint8_t *out; size_t len; } sink;
static void emit_run(sink *s, unsigned count, uint8_t v) { s->out[s->len++] = (uint8_t)count; s->out[s->len++] = v; }

/* BI_RLE8 the way bitmap.c:495-633 writes it, so that the files are equal byte for byte.  The coder is in one of three states
   per byte: IDLE (nothing pending), SAME (a run of equal bytes) or MIXED (a stretch whose neighbours differ).  A stretch ends at
   the picture row's end, at the 255th byte, or where the byte kind changes; SAME goes out as (count, value), MIXED as (0, count,
   bytes...) when it holds three or more bytes and as (1, value) pairs when it holds one or two.  The reference's quirks, kept:
     * its counter is eight bits wide, so a MIXED stretch that reaches 255 bytes wraps to 0 and leaves as 255 (1, value) pairs;
     * a MIXED stretch that is ended by the row end or by the 255th byte makes the reference step its read position twice:
       one byte of the picture is skipped and every later row is cut one byte late (the picture shears; such files need 255
       neighbouring columns, no two alike, or a row that ends on a lone byte -- the margins of this picture are white, so only the
       first happens, with thousands of scattered MEMs); bytes it then reads beyond the picture are taken as 0 here;
     * when the coded bytes reach the size of the plain picture, the reference gives up and writes the picture plain.
   Returns the coded length, or 0 for "write it plain". */
static size_t rle8_encode(const uint8_t *px, size_t n, size_t row, uint8_t *out) {
    enum { IDLE, SAME, MIXED } state = IDLE;
    sink s = {out, 0};
    size_t at = 0, mixed_from = 0, step;
    uint8_t pending = 0; /* bytes of the stretch in hand, modulo 256 as in the reference */
#define PX(i) ((i) < n ? px[(i)] : (uint8_t)0)
    for (step = 1; step <= n; step++) {
        uint8_t v = PX(at);
        int next_same = step != n && PX(at + 1) == v;
        int cut;
        pending++;
        cut = step == n || pending == 255 || step % row == 0;
        if (cut) {
            if (state == IDLE) state = SAME;
            if (state == SAME) next_same = 0;
            else { next_same = 1; at++; pending++; } /* MIXED: the byte in hand goes out with the stretch */
        }
        if (state == IDLE) {
            state = next_same ? SAME : MIXED;
            if (state == MIXED) mixed_from = at;
            at++;
            continue;
        }
        if (state == SAME) {
            if (!next_same) { emit_run(&s, pending, v); state = IDLE; pending = 0; }
        } else if (next_same) { /* MIXED ends before the byte in hand, which opens a run */
            uint8_t held = (uint8_t)(pending - 1);
            if (pending <= 3) {
                size_t k;
                for (k = mixed_from; k < at; k++) emit_run(&s, 1, PX(k));
            } else {
                size_t k;
                emit_run(&s, 0, held);
                for (k = mixed_from; k < at; k++) s.out[s.len++] = PX(k);
                if (held & 1) s.out[s.len++] = 0; /* literal stretches are padded to an even count */
            }
            state = cut ? IDLE : SAME;
            pending = cut ? 0 : 1;
        }
        at++;
        if (s.len >= n) return 0;
        if (step % row == 0) emit_run(&s, 0, 0); /* end of row */
    }
#undef PX
    emit_run(&s, 0, 1); /* end of picture */
    while (s.len % 4) s.out[s.len++] = 0;
    return s.len;
}

static void put16(uint8_t *p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { put16(p, v & 0xFFFFu); put16(p + 2, v >> 16); }

static int canvas_save(const canvas *cv, const char *path) {
    size_t plain = (size_t)cv->w * (size_t)cv->h, coded;
    uint8_t head[14 + 40 + 4 * NUM_COLOURS], *out;
    FILE *f;
    int i, ok;
    while (plain % 4) plain++;
    /* worst case of the coder: two bytes per byte (pairs) plus the row and picture ends */
    out = (uint8_t *)malloc(2 * plain + 2 * (size_t)cv->h + 16);
    if (!out) return 0;
    if (!(f = fopen(path, "wb"))) { free(out); return 0; }
    coded = rle8_encode(cv->px, plain, (size_t)cv->w, out);
    memset(head, 0, sizeof head);
    head[0] = 'B'; head[1] = 'M';
    put32(head + 2, (uint32_t)(sizeof head + (coded ? coded : plain)));
    put32(head + 10, (uint32_t)sizeof head);
    put32(head + 14, 40);
    put32(head + 18, (uint32_t)cv->w);
    put32(head + 22, (uint32_t)cv->h);
    put16(head + 26, 1);
    put16(head + 28, 8);
    put32(head + 30, coded ? 1 : 0); /* BI_RLE8 / BI_RGB */
    put32(head + 34, (uint32_t)(coded ? coded : plain));
    put32(head + 38, 1024);
    put32(head + 42, 1024);
    put32(head + 46, NUM_COLOURS);
    put32(head + 50, NUM_COLOURS);
    for (i = 0; i < NUM_COLOURS; i++) {
        head[54 + 4 * i + 0] = cv->rgb[i][2];
        head[54 + 4 * i + 1] = cv->rgb[i][1];
        head[54 + 4 * i + 2] = cv->rgb[i][0];
    }
    ok = fwrite(head, 1, sizeof head, f) == sizeof head;
    ok = fwrite(coded ? out : cv->px, 1, coded ? coded : plain, f) == (coded ? coded : plain) && ok;
    free(out);
    return (fclose(f) != EOF) && ok;
}

/* Test entry (libslamem_host.so): any 8-bit picture through the same writer -- rows top first, palette as in the tool.  Lets the
   tests reach what the MEM map never produces (a picture that does not compress: the plain fall-back of canvas_save). */
int slh_write_bmp8(const char *path, int width, int height, const uint8_t *rows_top_first) {
    canvas cv;
    int y, ok;
    if (!path || width <= 0 || height <= 0 || !rows_top_first) return 0;
    cv.w = width; cv.h = height;
    cv.px = (uint8_t *)malloc((size_t)width * (size_t)height + 4);
    if (!cv.px) return 0;
    memset(cv.px + (size_t)width * (size_t)height, 0, 4);
    palette_init(&cv);
    for (y = 0; y < height; y++)
        memcpy(cv.px + (size_t)(height - 1 - y) * (size_t)width, rows_top_first + (size_t)y * (size_t)width, (size_t)width);
    ok = canvas_save(&cv, path);
    free(cv.px);
    return ok;
}

/* ---- the MEMs file (slamem.c:387-439) -------------------------------------------------------------------------------------- */

typedef struct { const char *p, *end; } reader;
static int is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
static void skip_space(reader *r) { while (r->p < r->end && is_space(*r->p)) r->p++; }

static int read_int(reader *r, int *out) { /* what scanf's %d takes: blanks, a sign, digits */
    long long v = 0;
    int neg = 0, any = 0;
    skip_space(r);
    if (r->p < r->end && (*r->p == '-' || *r->p == '+')) { neg = *r->p == '-'; r->p++; }
    for (; r->p < r->end && *r->p >= '0' && *r->p <= '9'; r->p++, any = 1)
        if (v < (1LL << 40)) v = v * 10 + (*r->p - '0');
    if (!any) return 0;
    *out = (int)(neg ? -v : v);
    return 1;
}

static int ends_in_reverse(const char *name) { /* the strand marker slamem.c:102 writes; a name must be longer than it */
    size_t n = strlen(name);
    return n > 7 && strcmp(name + n - 7, "Reverse") == 0;
}

int slh_mem_map_image(const char *mems_path, const slh_record *seqs, int num_seqs, int num_refs, FILE *log) {
    picture pic;
    reader r;
    char *data = NULL, name[256], *image_path;
    long bytes = 0;
    int *sizes, i, track = 0, strand = 0, mems = 0, status = -1, at_start = 1;
    FILE *f;

    if (num_refs > 1) { /* slamem.c:362-366 */
        fprintf(log, "\n> ERROR: Support for visualizing multiple reference sequences is not implemented yet.\n");
        fprintf(log, "\tIf you require this feature, please request it to the author.\n");
        return -1;
    }
    fprintf(log, "> Processing MEMs from <%s> ...\n", mems_path);
    fflush(log);
    if (!(f = fopen(mems_path, "r"))) { fprintf(log, "\n> ERROR: Cannot read file\n"); return -1; }
    fseek(f, 0L, SEEK_END);
    bytes = ftell(f);
    rewind(f);
    data = (char *)malloc((size_t)(bytes > 0 ? bytes : 0) + 1);
    if (!data || (bytes > 0 && fread(data, 1, (size_t)bytes, f) != (size_t)bytes)) {
        fclose(f); free(data);
        fprintf(log, "\n> ERROR: Cannot read file\n");
        return -1;
    }
    fclose(f);
    sizes = (int *)malloc((size_t)num_seqs * sizeof(int));
    if (!sizes) { free(data); return -1; }
    for (i = 0; i < num_seqs; i++) sizes[i] = (int)seqs[i].size;
    if (picture_init(&pic, sizes, num_seqs) != 0) { fprintf(log, "\n> ERROR: Out of memory\n"); goto out; }

    r.p = data; r.end = data + (bytes > 0 ? bytes : 0);
    name[0] = '\0';
    for (;;) {
        int ref_pos, query_pos, len, fields;
        /* every item the reference reads ends by eating the blanks behind it, so only the file's first byte can be a blank
           when it looks for '>' -- and then the line is taken for a MEM (slamem.c:388-422) */
        int blank_first = at_start && r.p < r.end && is_space(*r.p);
        at_start = 0;
        skip_space(&r);
        if (r.p == r.end || (*r.p == '>' && !blank_first)) {
            size_t n = 0;
            if (track != 0) { fprintf(log, "(%d MEMs)\n", mems); fflush(log); }
            if (r.p == r.end) break;
            mems = 0;
            r.p++;
            skip_space(&r);
            while (r.p < r.end && *r.p != '\n' && n < 255) name[n++] = *r.p++;
            if (n > 0) name[n] = '\0'; /* an empty header leaves the previous name in place, as scanf does */
            fprintf(log, ":: '%s' ... ", name);
            fflush(log);
            if (ends_in_reverse(name)) { strand = 1; continue; }
            strand = 0;
            track++;
            if (track == num_seqs) {
                fprintf(log, "\n> ERROR: MEMs file not generated from this query file (too many sequences)\n");
                goto out;
            }
            while (track < num_seqs && strcmp(name, seqs[track].name) != 0) track++;
            if (track == num_seqs) { fprintf(log, "\n> ERROR: Sequence name was not found in query file\n"); goto out; }
            continue;
        }
        fields = read_int(&r, &ref_pos) && read_int(&r, &query_pos) && read_int(&r, &len);
        if (!fields || track == 0) { fprintf(log, "\n> ERROR: Invalid MEM format\n"); goto out; }
        if (ref_pos == 0 || query_pos == 0 || len == 0) { fprintf(log, "\n> ERROR: Invalid MEM values\n"); goto out; }
        ref_pos--; query_pos--; /* the file is 1-based */
        if (strand) query_pos = sizes[track] - (query_pos + len); /* reverse-strand positions count from the other end */
        picture_block(&pic, track, query_pos, ref_pos, len, strand);
        mems++;
    }
    picture_names(&pic, seqs);
    image_path = slh_append_to_basename(mems_path, ".bmp");
    fprintf(log, "> Saving image to <%s> ... ", image_path);
    fflush(log);
    if (!canvas_save(&pic.cv, image_path)) {
        fprintf(log, "\n> ERROR: Cannot write file\n");
        free(image_path);
        status = 0; /* the reference leaves with exit(0) here (graphics.c:374-377) */
        goto out;
    }
    fprintf(log, "OK\n");
    free(image_path);
    fprintf(log, "> Done!\n");
    status = 0;
out:
    picture_free(&pic);
    free(sizes);
    free(data);
    return status;
}
