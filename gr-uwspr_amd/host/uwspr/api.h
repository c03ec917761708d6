/* Export macro, as include/uwspr/api.h:27-31 of the reference. */
#ifndef INCLUDED_UWSPR_API_H
#define INCLUDED_UWSPR_API_H
#define UWSPR_API __attribute__((visibility("default")))
#endif
