/* Asks the installed libjpeg which JPEG_LIB_VERSION it was built as and how large it believes struct jpeg_decompress_struct to be: both come
 * back in the error it raises for a deliberately wrong call (jpeg_CreateDecompress checks `version` first, then `structsize`, and puts the
 * expected value into err->msg_parm.i[0]).  Prints "<version> <size>".  TEST INFRASTRUCTURE ONLY (oracle/build_io.py). */
#include <setjmp.h>
#include <stdio.h>
#include <string.h>

struct err_mgr { void (*error_exit)(void*); void (*f1)(void*, int); void (*f2)(void*); void (*f3)(void*, char*); void (*f4)(void*);
                 int msg_code; union { int i[8]; char s[80]; } msg_parm; char rest[1024]; };
struct err_mgr* jpeg_std_error(struct err_mgr*);
void jpeg_CreateDecompress(void* cinfo, int version, size_t structsize);

static jmp_buf env;
static void on_error(void* cinfo) { (void)cinfo; longjmp(env, 1); }

int main(void) {
    static struct err_mgr em;
    static unsigned char cinfo[8192];
    int version = -1, size = -1;
    for (int pass = 0; pass < 2; ++pass) {
        memset(cinfo, 0, sizeof(cinfo));
        jpeg_std_error(&em);
        em.error_exit = on_error;
        *(struct err_mgr**)cinfo = &em;
        if (setjmp(env) == 0) {
            jpeg_CreateDecompress(cinfo, pass == 0 ? -12345 : version, 1);
            return 2;   /* accepted a nonsense call: not the library this probe was written for */
        }
        if (pass == 0) version = em.msg_parm.i[0];   /* JERR_BAD_LIB_VERSION: expected, given */
        else size = em.msg_parm.i[0];                /* JERR_BAD_STRUCT_SIZE: expected, given */
    }
    printf("%d %d\n", version, size);
    return (version > 0 && size > 144) ? 0 : 1;
}
