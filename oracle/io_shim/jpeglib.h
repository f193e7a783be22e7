/* jpeglib.h stand-in for the pin harness (oracle/build_io.py).  TEST INFRASTRUCTURE ONLY.
 *
 * The image ships libjpeg.so.8 (libjpeg-turbo) but not its headers.  The reference's GUI/Tools/JPEGLoader.h uses the classic IJG calling
 * sequence (jpeg_std_error, jpeg_create_decompress, a hand-made jpeg_source_mgr, jpeg_read_header, jpeg_calc_output_dimensions,
 * jpeg_start_decompress, jpeg_read_scanlines, jpeg_finish_decompress, jpeg_destroy_decompress) and touches five fields of
 * jpeg_decompress_struct: err, mem (-> alloc_sarray), src, output_width, output_height.  This header declares exactly that API.  The
 * struct's leading fields are spelled out (they have not moved since IJG release 6b: the v7 / v8 ABI differences come further down) and the
 * rest is an opaque tail whose length comes from the LIBRARY: jpeg_CreateDecompress refuses a structure of the wrong size and reports the
 * size and version it expects -- oracle/build_io.py asks it (oracle/io_shim/jpeg_probe.c) and passes both in as
 * MFIO_JPEG_LIB_VERSION / MFIO_JPEG_DECOMPRESS_SIZE.  A wrong leading layout would show as a wrong image size or a crash in the pin test,
 * which decodes a real JPEG frame through this path and compares it with an independent decoder.
 */
#ifndef MFIO_JPEGLIB_H
#define MFIO_JPEGLIB_H
#include <stddef.h>

#ifndef MFIO_JPEG_LIB_VERSION
#error "MFIO_JPEG_LIB_VERSION / MFIO_JPEG_DECOMPRESS_SIZE come from oracle/io_shim/jpeg_probe.c (see oracle/build_io.py)"
#endif
#define JPEG_LIB_VERSION MFIO_JPEG_LIB_VERSION

typedef unsigned char JSAMPLE;
typedef JSAMPLE* JSAMPROW;
typedef JSAMPROW* JSAMPARRAY;
typedef unsigned char JOCTET;
typedef unsigned int JDIMENSION;
typedef int boolean;
#ifndef TRUE
#define TRUE 1
#endif
#ifndef FALSE
#define FALSE 0
#endif
#define JPOOL_PERMANENT 0
#define JPOOL_IMAGE 1
#define JMSG_STR_PARM_MAX 80

struct jpeg_common_struct;
struct jpeg_decompress_struct;
typedef struct jpeg_common_struct* j_common_ptr;
typedef struct jpeg_decompress_struct* j_decompress_ptr;

struct jpeg_error_mgr {
    void (*error_exit)(j_common_ptr cinfo);
    void (*emit_message)(j_common_ptr cinfo, int msg_level);
    void (*output_message)(j_common_ptr cinfo);
    void (*format_message)(j_common_ptr cinfo, char* buffer);
    void (*reset_error_mgr)(j_common_ptr cinfo);
    int msg_code;
    union { int i[8]; char s[JMSG_STR_PARM_MAX]; } msg_parm;
    int trace_level;
    long num_warnings;
    const char* const* jpeg_message_table;
    int last_jpeg_message;
    const char* const* addon_message_table;
    int first_addon_message;
    int last_addon_message;
    unsigned char mfio_slack[256];   /* the caller allocates this object: room to spare if a build of the library ever grew it */
};

struct jpeg_memory_mgr {
    void* (*alloc_small)(j_common_ptr cinfo, int pool_id, size_t sizeofobject);
    void* (*alloc_large)(j_common_ptr cinfo, int pool_id, size_t sizeofobject);
    JSAMPARRAY (*alloc_sarray)(j_common_ptr cinfo, int pool_id, JDIMENSION samplesperrow, JDIMENSION numrows);
    /* ... the library owns the object; nothing behind alloc_sarray is used here */
};

struct jpeg_source_mgr {
    const JOCTET* next_input_byte;
    size_t bytes_in_buffer;
    void (*init_source)(j_decompress_ptr cinfo);
    boolean (*fill_input_buffer)(j_decompress_ptr cinfo);
    void (*skip_input_data)(j_decompress_ptr cinfo, long num_bytes);
    boolean (*resync_to_restart)(j_decompress_ptr cinfo, int desired);
    void (*term_source)(j_decompress_ptr cinfo);
};

struct jpeg_common_struct {
    struct jpeg_error_mgr* err; struct jpeg_memory_mgr* mem; void* progress; void* client_data; boolean is_decompressor; int global_state;
};

struct jpeg_decompress_struct {
    struct jpeg_error_mgr* err; struct jpeg_memory_mgr* mem; void* progress; void* client_data; boolean is_decompressor; int global_state;
    struct jpeg_source_mgr* src;
    JDIMENSION image_width, image_height;
    int num_components;
    int jpeg_color_space;
    int out_color_space;
    unsigned int scale_num, scale_denom;
    double output_gamma;
    boolean buffered_image, raw_data_out;
    int dct_method;
    boolean do_fancy_upsampling, do_block_smoothing, quantize_colors;
    int dither_mode;
    boolean two_pass_quantize;
    int desired_number_of_colors;
    boolean enable_1pass_quant, enable_external_quant, enable_2pass_quant;
    JDIMENSION output_width, output_height;
    unsigned char mfio_opaque[MFIO_JPEG_DECOMPRESS_SIZE - 144];
};

#ifdef __cplusplus
extern "C" {
#endif
struct jpeg_error_mgr* jpeg_std_error(struct jpeg_error_mgr* err);
void jpeg_CreateDecompress(j_decompress_ptr cinfo, int version, size_t structsize);
#define jpeg_create_decompress(cinfo) jpeg_CreateDecompress((cinfo), JPEG_LIB_VERSION, (size_t)sizeof(struct jpeg_decompress_struct))
int jpeg_read_header(j_decompress_ptr cinfo, boolean require_image);
void jpeg_calc_output_dimensions(j_decompress_ptr cinfo);
boolean jpeg_start_decompress(j_decompress_ptr cinfo);
JDIMENSION jpeg_read_scanlines(j_decompress_ptr cinfo, JSAMPARRAY scanlines, JDIMENSION max_lines);
boolean jpeg_finish_decompress(j_decompress_ptr cinfo);
void jpeg_destroy_decompress(j_decompress_ptr cinfo);
boolean jpeg_resync_to_restart(j_decompress_ptr cinfo, int desired);
#ifdef __cplusplus
}
#endif
#endif /* MFIO_JPEGLIB_H */
