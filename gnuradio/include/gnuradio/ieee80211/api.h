/* symbol visibility of libgnuradio-ieee80211 (the usual OOT api.h) */
#ifndef INCLUDED_IEEE80211_API_H
#define INCLUDED_IEEE80211_API_H

#include <gnuradio/attributes.h>

#ifdef gnuradio_ieee80211_EXPORTS
#define IEEE80211_API __GR_ATTR_EXPORT
#else
#define IEEE80211_API __GR_ATTR_IMPORT
#endif

#endif /* INCLUDED_IEEE80211_API_H */
